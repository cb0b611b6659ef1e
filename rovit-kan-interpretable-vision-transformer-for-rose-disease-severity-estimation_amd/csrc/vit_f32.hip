// fp32 reference-precision forward of the DeiT-Tiny backbone (inference only).
//
// Why it exists: BASELINE.json's north_star asks for logits/severity within 1e-3 of the reference's CPU path in fp32 and
// a bit-exact class argmax.  The training path computes its GEMMs with bf16 operands (the precision class of the
// reference's own CUDA autocast path), which moves the features by ~5e-3 RMS -- enough to carry a feature across the
// discontinuity of the reference's truncated spline.  This path runs the SAME arithmetic (timm VisionTransformer.forward,
// SURVEY.md section 2, reached from /root/reference/models/backbone.py:12-25) entirely in fp32 on the GPU, so the end-to-end
// parity statement can be made at fp32 tolerance.  Round 4: the GEMMs and the two attention products run on the fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: fp32 operands, every product and sum an exact fp32 FMA chain in a fixed k order, no bf16
// anywhere -- MI355X_MICROARCH.md "FP32-input MFMA"), 157 TFLOP/s peak = 1/16 of the bf16 rate: 6.90 ms per 256 images = 93 TFLOP/s (the
// round-2 VALU tiles ran at 41 TFLOP/s, 15.5 ms).
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int T = 197, D = 192, H = 3, HD = 64, MLP = 768, PD = 768;
enum { P_CLS = 0, P_POS, P_PATCH_W, P_PATCH_B, P_NORM_W, P_NORM_B, P_BLOCK0 };
enum { B_N1W = 0, B_N1B, B_QKVW, B_QKVB, B_PROJW, B_PROJB, B_N2W, B_N2B, B_FC1W, B_FC1B, B_FC2W, B_FC2B, B_COUNT };
enum { F_NONE = 0, F_GELU = 1, F_RESID = 2, F_PATCH = 3 };

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// C[M,N] = A[M,K] W[N,K]^T + bias on the fp32 matrix cores.  Workgroup = 4 WN waves = a 128 x (32 NT WN) output tile, wave (wm, wn) =
// rows 32 wm .. + 31 x NT accumulator tiles of 32 x 32.  The library runs WN = 1, NT = 2: a 128 x 64 tile of 4 waves, five workgroups per CU
// (27.6 KB of LDS, 88-96 VGPRs) -- small workgroups in different phases hide each other's loads, barriers and epilogues, and the tile count
// no longer leaves a third of the CUs with twice the work (tools/lab/f32_gemm_lab.hip times it against WN = 2, NT = 3, the 128 x 192 tile
// of 8 waves it replaced: qkv 140 -> 114 us, fc1 170 -> 143, fc2 175 -> 157 per 256 images, same bits).
// K walks in steps of 32 through ONE LDS stage (A 128 x 32 and W BN x 32 floats, rows padded to 36 floats: ds_read_b128 of 16
// rows x 16 bytes is conflict-free); the next stage is requested from global memory before the stage is computed and written behind it.
// Operand order inside a 32-deep stage: a lane holds 4 consecutive k of its row from one 16-byte read and feeds them to 4
// MFMAs; the lane half h = lane >> 5 takes k = 8c + 4h + s in MFMA s of chunk c -- the SAME map for A and for W, so the products pair
// up correctly and the summation order (k = 8c + s, then 8c + 4 + s, for s, then c, then stage) is fixed: bit-reproducible.
//   F_GELU : exact-erf GELU on the result          F_RESID: C += result (in place on the residual stream)
//   F_PATCH: A is gathered from the NCHW image (row m = image b, patch p; k = c*256 + py*16 + px), the result goes to token
//            row b*T + 1 + p with the position embedding added (timm PatchEmbed + pos_embed)
constexpr int GBM = 128, GBK = 32, GST = GBK + 4;                // GST: LDS row stride in floats
// (LAB: ablation bits of tools/lab/f32_gemm_lab.hip -- 1 no global loads inside the loop, 2 no restaging at all, 4 no epilogue, 8 the MFMAs
// alone on register operands; the library instantiates LAB = 0 only)
// LayerNorm without a launch of its own (the model's norm1 / norm2 sit between a GEMM that completes the residual rows and a GEMM that reads
// their normalised copy):
//   STO: the epilogue of the GEMM that writes the rows (F_RESID, F_PATCH; N = 192 = three 64-column tiles) also writes, per row and column
//        tile, the sum and the sum of squares of its 64 values: stats[row][ct][2] (a halving butterfly over the 32 lanes of a lane half; every
//        slot has one writer, so the statistics are the same bits run to run);
//   LNA: the GEMM that consumes the normalised rows (F_NONE = qkv, F_GELU = fc1; K = 192) reads the raw rows, folds the three partial sums in
//        a fixed order (mean = sum / 192, var = sum of squares / 192 - mean^2, clamped at 0) and applies ((x - mean) rstd) gamma + beta -- the
//        element-wise order of ln_f32_kernel -- on the staging registers before they go to LDS; gamma / beta and the tile's (mean, rstd) wait in
//        LDS (2.5 KB).
// One-pass variance in fp32 is good to ~2e-7 (1 + mean^2 / var): fine for the model's rows (|mean| below the spread), not for a row that is a
// large offset plus a small signal.  The consumer therefore checks mean^2 > 64 var per row and, for such a row only, recomputes both moments
// from the row itself in two passes with fp64 accumulators (192 loads by one thread: never taken on the model's own activations, ~2 us per flagged tile when it
// is; tests: a backbone with its position embedding shifted by +300).  Carrying the sums in fp64 instead was measured: 6.87 -> 6.95 ms per 256
// images for the 64-bit butterfly.  (Whole model against the CPU oracle: features 3.6e-6, as before the fold.)
struct F32Ln {
  const float* stats_in;      // LNA
  const float* gamma;
  const float* beta;
  float* stats_out;           // STO
  float eps;
};
template <int EPI, int LAB = 0, int WN = 2, int NT = 3, bool LNA = false, bool STO = false>
__global__ __launch_bounds__(256 * WN, WN == 2 ? 2 : 5) void gemm_f32_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N,
                                                              int K, const float* __restrict__ pos, const F32Ln ln) {
  constexpr int BN = 32 * NT * WN, SR = 32 * WN, QA = GBM / SR, QW = BN / SR;      // SR: rows one staging step of the workgroup covers
  __shared__ __attribute__((aligned(16))) float As[GBM * GST];
  __shared__ __attribute__((aligned(16))) float Ws[BN * GST];
  __shared__ __attribute__((aligned(16))) float Gs[LNA ? 2 * D : 4];               // gamma[192], beta[192]
  __shared__ __attribute__((aligned(16))) float2 Ms[LNA ? GBM : 1];                // (mean, rstd) of the tile's rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (1-D grid; workgroup id L runs on XCD L % 8): the N / BN column tiles of a row tile read the same A rows, so they are
  // given to ONE XCD, back to back -- its L2 fetches the A tile once instead of every XCD fetching it.  Row tile = 8 (slot / ncol) + xcd.
  const int ncol = N / BN, nrow = (M + GBM - 1) / GBM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = (slot / ncol) * 8 + xcd, ct = slot - (slot / ncol) * ncol;
  if (rt >= nrow) return;                                          // (the grid is padded to a multiple of 8 row tiles)
  const int m0 = rt * GBM, n0 = ct * BN;
  // staging: thread t moves 16 bytes of row t / 8 (+ SR per step), chunk t % 8; 8 waves: 2 steps for A (128 rows), 3 for W (192 rows)
  const int srow = tid >> 3, sch = tid & 7;
  const float* a_src[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    int m = m0 + srow + SR * q;
    m = m < M ? m : M - 1;                                         // clamped: rows beyond M are computed and never stored
    if (EPI == F_PATCH) {
      const int b = m / (T - 1), pch = m - b * (T - 1);
      a_src[q] = A + ((size_t)b * 3 * 224 + (pch / 14) * 16) * 224 + (pch % 14) * 16;     // + c*224*224 + py*224 + px
    } else {
      a_src[q] = A + (size_t)m * lda;
    }
  }
  if (LNA) {                                                       // (both visible behind the loop's first barrier)
    for (int i = tid; i < 2 * D; i += 256 * WN) Gs[i] = i < D ? ln.gamma[i] : ln.beta[i - D];
    if (tid < GBM) {
      int m = m0 + tid;
      m = m < M ? m : M - 1;
      const float* sp = ln.stats_in + (size_t)m * 6;
      float mean = ((sp[0] + sp[2]) + sp[4]) * (1.f / D);
      float var = fmaxf(((sp[1] + sp[3]) + sp[5]) * (1.f / D) - mean * mean, 0.f);
      if (mean * mean > 64.f * var) {                                // ill-conditioned for the one-pass form: two passes over the row itself
        const float* xr = A + (size_t)m * lda;                       // (fp64 accumulators: the path is rare, its accuracy is the point)
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < D; ++k) s1 += (double)xr[k];
        const double mu = s1 * (1.0 / D);
        for (int k = 0; k < D; ++k) { const double d = (double)xr[k] - mu; s2 += d * d; }
        mean = (float)mu;
        var = (float)(s2 * (1.0 / D));
      }
      Ms[tid] = make_float2(mean, 1.f / sqrtf(var + ln.eps));
    }
  }
  auto load_a = [&](int q, int k0) -> float4 {
    const int k = k0 + 4 * sch;
    if (EPI == F_PATCH) return *(const float4*)(a_src[q] + ((size_t)(k >> 8) * 224 + ((k >> 4) & 15)) * 224 + (k & 15));
    return *(const float4*)(a_src[q] + k);
  };
  auto load_w = [&](int q, int k0) -> float4 { return *(const float4*)(W + (size_t)(n0 + srow + SR * q) * K + k0 + 4 * sch); };
  float4 pa[QA], pw[QW];
#pragma unroll
  for (int q = 0; q < QA; ++q) pa[q] = load_a(q, 0);
#pragma unroll
  for (int q = 0; q < QW; ++q) pw[q] = load_w(q, 0);
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += GBK) {
    if (!(LAB & 2) || k0 == 0) {
    __syncthreads();                                               // every wave is done reading the previous stage
    if (LNA) {
      const float4 g4 = *(const float4*)&Gs[k0 + 4 * sch], b4 = *(const float4*)&Gs[D + k0 + 4 * sch];
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        const float2 mr = Ms[srow + SR * q];
        pa[q].x = (pa[q].x - mr.x) * mr.y * g4.x + b4.x;
        pa[q].y = (pa[q].y - mr.x) * mr.y * g4.y + b4.y;
        pa[q].z = (pa[q].z - mr.x) * mr.y * g4.z + b4.z;
        pa[q].w = (pa[q].w - mr.x) * mr.y * g4.w + b4.w;
      }
    }
#pragma unroll
    for (int q = 0; q < QA; ++q) *(float4*)&As[(srow + SR * q) * GST + 4 * sch] = pa[q];
#pragma unroll
    for (int q = 0; q < QW; ++q) *(float4*)&Ws[(srow + SR * q) * GST + 4 * sch] = pw[q];
    __syncthreads();
    }
    if (!(LAB & 3) && k0 + GBK < K) {                              // next stage: in flight while this one is computed
#pragma unroll
      for (int q = 0; q < QA; ++q) pa[q] = load_a(q, k0 + GBK);
#pragma unroll
      for (int q = 0; q < QW; ++q) pw[q] = load_w(q, k0 + GBK);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 a4, b4[NT];
      if ((LAB & 8) && k0 > 0) {                                   // (lab: the matrix instructions alone, on whatever the registers hold)
        a4 = pa[0];
#pragma unroll
        for (int t = 0; t < NT; ++t) b4[t] = pw[t % QW];
      } else {
        a4 = *(const float4*)&As[(32 * wm + l31) * GST + 8 * c + 4 * lh];
#pragma unroll
        for (int t = 0; t < NT; ++t) b4[t] = *(const float4*)&Ws[(32 * NT * wn + 32 * t + l31) * GST + 8 * c + 4 * lh];
      }
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv[4] = {b4[t].x, b4[t].y, b4[t].z, b4[t].w};
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sidx], bv[sidx], acc[t], 0, 0, 0);
        }
    }
  }
  // accumulator tile: column n = lane & 31, row m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): a register is two 128-byte row segments
  float bn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bn[t] = bias ? bias[n0 + 32 * NT * wn + 32 * t + l31] : 0.f;
  auto out_row = [&](int m) -> size_t {                             // output row of tile row m (rows beyond M: computed, never stored)
    size_t orow = (size_t)(m < M ? m : M - 1);
    if (EPI == F_PATCH) {
      const int b = (int)orow / (T - 1), pch = (int)orow - b * (T - 1);
      orow = (size_t)b * T + 1 + pch;
    }
    return orow;
  };
  float st[STO ? 32 : 1];                                           // st[2 r + j]: this lane's part of row r's sum (j = 0) / sum of squares (1)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
    float s1 = 0.f, s2 = 0.f;
    const size_t orow = out_row(m);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 32 * NT * wn + 32 * t + l31;
      if ((LAB & 4) && acc[t][r] != 12345.678f) continue;
      float v = acc[t][r] + bn[t];
      if (EPI == F_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));      // exact-erf GELU (timm default)
      if (EPI == F_PATCH) v += pos[(orow % T) * N + n];
      if (EPI == F_RESID) v += C[orow * ldc + n];
      if (m < M) C[orow * ldc + n] = v;
      s1 += v;
      s2 += v * v;
    }
    if (STO) { st[2 * r] = s1; st[2 * r + 1] = s2; }
  }
  if (STO) {
    // 32 values per lane, to be summed over the 32 lanes of a lane half (the wave holds the rows' 64 values of column tile ct: WN = 1):
    // halving butterfly -- at distance o a lane keeps the half of its values whose index has bit o set like its own lane number and adds
    // the partner's copy of that half; 16 + 8 + 4 + 2 + 1 exchanges instead of 5 x 32, and lane l ends with the total of value l.
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
      const bool up = (l31 & o) != 0;
#pragma unroll
      for (int i = 0; i < o; ++i) {
        const float keep = up ? st[i + o] : st[i], send = up ? st[i] : st[i + o];
        st[i] = keep + __shfl_xor(send, o);
      }
    }
    const int r = l31 >> 1, m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < M) ln.stats_out[out_row(m) * 6 + ct * 2 + (l31 & 1)] = st[0];
  }
}

// LayerNorm with affine over rows of 192 floats: one wave per row; row r of the output comes from row r * row_step of x
__global__ __launch_bounds__(256) void ln_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, int rows, int row_step, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * row_step * D;
  float v[3], s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { v[i] = xr[lane + 64 * i]; s += v[i]; }
  const float mean = wave_sum64(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { v[i] -= mean; q += v[i] * v[i]; }
  const float rstd = 1.f / sqrtf(wave_sum64(q) * (1.f / D) + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) y[(size_t)row * D + lane + 64 * i] = v[i] * rstd * gamma[lane + 64 * i] + beta[lane + 64 * i];
}

// softmax(q k^T * scale) v on the fp32 matrix cores.  Workgroup = 4 waves = HALF of an (image, head): wave w of half g owns queries
// 32 (4 g + w) .. + 31 (seven query tiles: half 0 has four, half 1 three and an idle wave).  ONE LDS tile (224 zero-padded rows x 64 fp32, rows
// padded to 68 floats; 60.9 KB) holds K for the first product and V for the second, so TWO workgroups share a CU and one's loads, softmax
// and output run beside the other's matrix phases (a workgroup per (image, head) with K and V both resident -- 7 waves, 106 KB, one per
// CU, every phase exposed -- took 113 us per 256-image block; the halves read K and V twice from L2, 2 x 39 MB).
//   S^T[key][q] = K Q^T: A = K rows from LDS (one 16-byte read feeds 4 MFMAs), B = the wave's scaled Q rows in 32 registers (lane half h
//   holds d = 32h + s for step s -- the same map on the K side); seven 32 x 32 accumulator tiles = the whole 224-key column block.
//   softmax down the accumulator registers of a lane (+ one exchange with lane ^ 32), exact expf, masked beyond T.
//   O^T[d][q] = V^T P^T: the probabilities ARE the B operand as they stand -- register r of tile kt holds key 32kt + (r&3) + 8(r>>2) + 4h,
//   so step r contracts over exactly those two keys and the A operand is V[that key][32 dt + lane & 31] (a 128-byte row read per half).
// Output through a wave-private LDS patch (in the tile, free behind a barrier) so that rows leave as 256-byte runs.
// (Three workgroups per CU -- a 200-row tile, 54.4 KB -- need the kernel under 168 VGPRs: 152 bytes of scratch, 126 us instead of 105.)
constexpr int AKS = HD + 4, ATP = 224, ANW = 4, ACH = ATP * 16 / (ANW * 64);       // ACH: 16-byte chunks of the tile per thread (14)
constexpr size_t ATTN_LDS = (size_t)ATP * AKS * sizeof(float);
__global__ __launch_bounds__(ANW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, float scale) {
  extern __shared__ __attribute__((aligned(16))) float KV[];
  const int bh = blockIdx.x >> 1, half = blockIdx.x & 1, b = bh / H, h = bh - b * H;
  const int ld = 3 * D;
  const float* base = qkv + (size_t)b * T * ld + h * HD;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_assert(ACH * ANW * 64 == ATP * 16, "whole chunks per thread");
  const int q0 = 32 * (4 * half + w);
  const bool active = q0 < T;                           // wave-uniform; the idle wave of half 1 only stages and keeps the barriers company
  float4 qraw[8];                                       // the wave's 32 query rows: lane half h holds d = 32h .. 32h + 31 of row q0 + lane & 31
  {
    const int qr = q0 + l31, qc = qr < T ? qr : T - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) qraw[i] = *(const float4*)(base + (size_t)qc * ld + 32 * lh + 4 * i);
  }
  // global -> registers -> LDS of K (col = D) or V (col = 2 D), rows beyond T zero; NB chunks per batch are in flight together
  auto stage = [&](int col) {
    constexpr int NB = ACH / 2;
#pragma unroll
    for (int i0 = 0; i0 < ACH; i0 += NB) {
      float4 reg[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + (i0 + i) * ANW * 64, r = e >> 4, c = e & 15, rc = r < T ? r : T - 1;
        reg[i] = *(const float4*)(base + (size_t)rc * ld + col + 4 * c);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = tid + (i0 + i) * ANW * 64, r = e >> 4, c = e & 15;
        if (r >= T) reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)&KV[r * AKS + 4 * c] = reg[i];
      }
    }
  };
  stage(D);
  float qv[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 t = qraw[i];
    qv[4 * i] = t.x * scale; qv[4 * i + 1] = t.y * scale; qv[4 * i + 2] = t.z * scale; qv[4 * i + 3] = t.w * scale;     // timm scales q first
  }
  __syncthreads();
  f32x16 st[7];
  float inv_l = 0.f;
  if (active) {
    // the K fragments of step (kt, c) + 2 are requested before the four MFMAs of step (kt, c) are issued (three register sets)
    const float* krow = &KV[l31 * AKS + 32 * lh];
    float4 kf[3];
    kf[0] = *(const float4*)krow;
    kf[1] = *(const float4*)(krow + 4);
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int step = 8 * kt + c, nx = step + 2;
        if (nx < 56) kf[nx % 3] = *(const float4*)(krow + 32 * (nx >> 3) * AKS + 4 * (nx & 7));
        __builtin_amdgcn_sched_barrier(0);
        const float4 k4 = kf[step % 3];
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qv[4 * c], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qv[4 * c + 1], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qv[4 * c + 2], st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qv[4 * c + 3], st[kt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (kt == 6 && key >= T) st[kt][r] = -INFINITY;
        m = fmaxf(m, st[kt][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(st[kt][r] - m);
        st[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32);
    inv_l = 1.f / l;
  }
  __syncthreads();                                      // every wave is done with K: V takes the tile
  stage(2 * D);
  __syncthreads();
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  if (active) {
    // contraction steps in key order: step s = 16 kt + r pairs keys 32 kt + (r & 3) + 8 (r >> 2) (+ 4 in the upper lane half); both keys of
    // steps 100 .. 111 are padding (T = 197).  The V operands of step s + 3 are requested before the two MFMAs of step s are issued.
    constexpr int NSTEP = 100;
    static_assert(T == 197, "step count of the P V product");
    const float* vbase = &KV[4 * lh * AKS + l31];
    float va[4], vb[4];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float* vrow = vbase + (32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2)) * AKS;
      va[s] = vrow[0]; vb[s] = vrow[32];
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int n = s + 3;
      if (n < NSTEP) {
        const float* vrow = vbase + (32 * (n >> 4) + (n & 3) + 8 * ((n & 15) >> 2)) * AKS;
        va[n & 3] = vrow[0]; vb[n & 3] = vrow[32];
      }
      __builtin_amdgcn_sched_barrier(0);
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s & 3], st[s >> 4][s & 15], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[s & 3], st[s >> 4][s & 15], o[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                      // nobody reads V any more: the tile becomes four 32 x 68 output patches
  float* patch = KV + w * 32 * AKS;
  if (active) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[l31 * AKS + 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * lh] = o[dt][r] * inv_l;
    // (same wave writes and reads the patch: the compiler's lgkmcnt wait orders them)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + (lane >> 4), qr = q0 + row;
      const float4 v = *(const float4*)&patch[row * AKS + 4 * (lane & 15)];
      if (qr < T) *(float4*)(out + ((size_t)b * T + qr) * D + h * HD + 4 * (lane & 15)) = v;
    }
  }
}

// the class-token row of every image (cls + pos[0]) and its LayerNorm statistics (slot 0 of the row's three; see F32Ln): one wave per image
__global__ __launch_bounds__(256) void cls_rows_f32_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ X,
                                                          float* __restrict__ stats, int B) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float v = cls[lane + 64 * i] + pos[lane + 64 * i];
    X[(size_t)b * T * D + lane + 64 * i] = v;
    s1 += v; s2 += v * v;
  }
  s1 = wave_sum64(s1); s2 = wave_sum64(s2);
  if (lane < 6) stats[(size_t)b * T * 6 + lane] = lane == 0 ? s1 : lane == 1 ? s2 : 0.f;
}

inline int gemm_f32_grid(int M, int N, int BN) { return ((M + GBM - 1) / GBM + 7) / 8 * 8 * (N / BN); }

template <int EPI, bool LNA = false, bool STO = false>
int gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, const float* pos, hipStream_t st,
             const F32Ln ln = F32Ln{nullptr, nullptr, nullptr, nullptr, 0.f}) {
  // 128 x 64 tiles of 4 waves, five workgroups per CU (tools/lab/f32_gemm_lab.hip, 256 images: qkv 140 -> 114 us, fc1 170 -> 143, fc2 with the
  // residual epilogue 175 -> 157, proj 62 -> 54 against the 128 x 192 tiles of 8 waves; same bits)
  constexpr int WN = 1, NT = 2, BN = 32 * NT * WN;
  ROVIT_CHECK_ARG(N % BN == 0 && K % GBK == 0, ROVIT_ERR_SHAPE, "gemm_f32: N must be a multiple of %d and K of %d (got %d, %d)", BN, GBK, N, K);
  ROVIT_CHECK_ARG((!LNA || (K == D && ln.stats_in && ln.gamma && ln.beta)) && (!STO || (N == D && ln.stats_out)), ROVIT_ERR_SHAPE,
                  "gemm_f32: the LayerNorm hooks need K = %d (consumer) / N = %d (producer)", D, D);
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 0, WN, NT, LNA, STO>), dim3(gemm_f32_grid(M, N, BN)), dim3(256 * WN), 0, st, A, lda, W, bias, C, ldc,
                     M, N, K, pos, ln);
  ROVIT_CHECK_LAUNCH("gemm_f32_mfma_kernel");
  return ROVIT_OK;
}

#define RUN(call) do { int rc__ = (call); if (rc__ != ROVIT_OK) return rc__; } while (0)
constexpr int F32_TWO_CHAINS_FROM = 192;           // batch from which the forward runs as two half-batch chains (see rovit_vit_forward_f32)

// the two events of the half-batch fork / join, one pair per device
struct ForkJoin { hipEvent_t fork = nullptr, join = nullptr; };
ForkJoin* fork_join() {
  static ForkJoin per_dev[64];
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  ForkJoin& f = per_dev[d];
  if (!f.fork && hipEventCreateWithFlags(&f.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (!f.join && hipEventCreateWithFlags(&f.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  return &f;
}

}  // namespace

// workspace: X (M,192) + row statistics (M,6; in a (M,192) slot) + qkv (M,576) + o (M,192) + h (M,768), all fp32, M = batch * 197
extern "C" size_t rovit_vit_f32_workspace_bytes(int batch) {
  const size_t M = (size_t)batch * T;
  return al(M * D * 4) * 3 + al(M * 3 * D * 4) + al(M * MLP * 4);
}

// images fp32 NCHW (B,3,224,224) -> features fp32 (B,192), every operation in fp32 (see the file header).  params as for
// rovit_vit_forward (the ORIGINAL fp32 parameters; no prepared weights).
// The images are independent, so the batch is cut into two halves that run the same launches on two HIP streams (the caller's and the
// device's side stream): the tile counts of one half's GEMM never fill the 512 workgroup slots evenly (fc2 of 256 images is 394 tiles:
// 138 CUs carry two and 118 one), and the other half's launches take what is left free.  Every result is bit-identical to the
// one-stream order -- a tile's arithmetic does not depend on which launch computes it.  Measured (tools/ab_f32_streams.sh, one box): 8.31 -> 7.8-8.0 ms
// at batch 256, but 4.97 -> 5.26 at 128 and 3.49 -> 4.44 at 64 (the half-sized grids no longer fill the chip): two chains from batch 192 up.
// (With the final 128 x 64-tile kernels: 7.09 -> 6.90 at 256; below 192 two chains are within +-1.5 % of one -- 2.60 / 3.27 / 4.10 / 4.68 ms at
// 64 / 96 / 128 / 160 on one chain, 2.48 / 3.32 / 4.05 / 4.74 on two -- so the threshold stays.)
extern "C" int rovit_vit_forward_f32(const float* images, const float* const* params, void* workspace, float* features, int batch, int depth,
                                     rovit_stream_t stream) {
  ROVIT_CHECK_ARG(images && params && workspace && features, ROVIT_ERR_NULL, "vit_forward_f32: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && depth > 0, ROVIT_ERR_SHAPE, "vit_forward_f32: bad batch/depth");
  hipStream_t st0 = (hipStream_t)stream;
  const int M = batch * T;
  char* ws = (char*)workspace;
  size_t o = 0;
  float* X = (float*)(ws + o); o += al((size_t)M * D * 4);
  float* stats = (float*)(ws + o); o += al((size_t)M * D * 4);      // (M, 3, 2) row statistics; the slot once held the normalised rows
  float* ao = (float*)(ws + o); o += al((size_t)M * D * 4);
  float* qkv = (float*)(ws + o); o += al((size_t)M * 3 * D * 4);
  float* hbuf = (float*)(ws + o);
  const float eps = 1e-6f;
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_f32_mfma_kernel, ATTN_LDS), ROVIT_ERR_LAUNCH, "vit_forward_f32: cannot raise the LDS limit");
  hipStream_t st1 = batch >= F32_TWO_CHAINS_FROM ? rovit_side_stream_handle() : nullptr;
  ForkJoin* fj = st1 ? fork_join() : nullptr;
  if (!fj) st1 = nullptr;
  struct Half { int b0, nb; hipStream_t st; };
  const Half halves[2] = {{0, st1 ? (batch + 1) / 2 : batch, st0}, {(batch + 1) / 2, batch / 2, st1}};
  const int nh = st1 ? 2 : 1;
  if (st1 && (hipEventRecord(fj->fork, st0) != hipSuccess || hipStreamWaitEvent(st1, fj->fork, 0) != hipSuccess)) {
    rovit_set_error("vit_forward_f32: event hand-over failed");
    return ROVIT_ERR_LAUNCH;
  }
  for (int hh = 0; hh < nh; ++hh) {
    const Half& h = halves[hh];
    hipStream_t st = h.st;
    const size_t r0 = (size_t)h.b0 * T;                   // first token row of the half
    const int Mh = h.nb * T;
    float *Xh = X + r0 * D, *sth = stats + r0 * 6, *aoh = ao + r0 * D, *qkvh = qkv + r0 * 3 * D, *hh_ = hbuf + r0 * MLP;
    const F32Ln sto{nullptr, nullptr, nullptr, sth, eps};           // producers: the row statistics go to sth
    hipLaunchKernelGGL(cls_rows_f32_kernel, dim3((h.nb + 3) / 4), dim3(256), 0, st, params[P_CLS], params[P_POS], Xh, sth, h.nb);
    ROVIT_CHECK_LAUNCH("cls_rows_f32_kernel");
    RUN((gemm_f32<F_PATCH, false, true>(images + (size_t)h.b0 * 3 * 224 * 224, 0, params[P_PATCH_W], params[P_PATCH_B], Xh, D, h.nb * (T - 1), D, PD,
                                        params[P_POS], st, sto)));
    for (int i = 0; i < depth; ++i) {
      const float* const* bp = params + P_BLOCK0 + B_COUNT * i;
      // norm1 / norm2 have no launch: statistics from the producer's epilogue, normalisation on the consumer's staging registers (F32Ln)
      RUN((gemm_f32<F_NONE, true>(Xh, D, bp[B_QKVW], bp[B_QKVB], qkvh, 3 * D, Mh, 3 * D, D, nullptr, st, F32Ln{sth, bp[B_N1W], bp[B_N1B], nullptr, eps})));
      hipLaunchKernelGGL(attn_f32_mfma_kernel, dim3(h.nb * H * 2), dim3(ANW * 64), ATTN_LDS, st, qkvh, aoh, 0.125f);
      ROVIT_CHECK_LAUNCH("attn_f32_mfma_kernel");
      RUN((gemm_f32<F_RESID, false, true>(aoh, D, bp[B_PROJW], bp[B_PROJB], Xh, D, Mh, D, D, nullptr, st, sto)));
      RUN((gemm_f32<F_GELU, true>(Xh, D, bp[B_FC1W], bp[B_FC1B], hh_, MLP, Mh, MLP, D, nullptr, st, F32Ln{sth, bp[B_N2W], bp[B_N2B], nullptr, eps})));
      RUN((gemm_f32<F_RESID, false, true>(hh_, MLP, bp[B_FC2W], bp[B_FC2B], Xh, D, Mh, D, MLP, nullptr, st, sto)));
    }
    // final LayerNorm on the class token of every image (row step T)
    hipLaunchKernelGGL(ln_f32_kernel, dim3((h.nb + 3) / 4), dim3(256), 0, st, Xh, params[P_NORM_W], params[P_NORM_B],
                       features + (size_t)h.b0 * D, h.nb, T, eps);
    ROVIT_CHECK_LAUNCH("ln_f32_kernel");
  }
  if (st1 && (hipEventRecord(fj->join, st1) != hipSuccess || hipStreamWaitEvent(st0, fj->join, 0) != hipSuccess)) {
    rovit_set_error("vit_forward_f32: event hand-over failed");
    return ROVIT_ERR_LAUNCH;
  }
  return ROVIT_OK;
}
