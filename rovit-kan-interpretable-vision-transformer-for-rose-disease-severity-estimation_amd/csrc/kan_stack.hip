// Fused KAN stack forward: every layer of KANSeverityModule (192 -> 64 -> 16 -> 1 by default) in ONE launch.
//
// Reference being restated: KANSeverityModule.forward (/root/reference/models/kan.py:138-149) = KANLayer.forward
// (:70-95, incl. BSplineBasis.compute_basis :8-44) -> ReLU -> ... -> KANLayer -> 3*sigmoid.
//
// One workgroup owns TB samples for the whole stack (TB x 8 threads; thread = (sample, group of outputs)):
//   * activations never leave the CU between layers (two small LDS buffers; they are ALSO written to HBM once,
//     because the backward and get_activation_trajectory need them: 81 floats per sample);
//   * the spline weights are used in a PREPARED layout Wt[feature][basis k][output o] (rovit_kan_prepare: one tiny
//     transposition per parameter update), so that the four basis rows j-3..j a (sample, feature) pair needs are each one
//     contiguous run over the outputs.  A layer is walked in chunks of IC input features: the chunk's slab is one
//     contiguous block, copied to LDS with 16-byte loads, and the contraction reads it with 16-byte LDS loads -- no
//     per-thread L2 gathers (the round-1 kernel gathered W[i, o, j-3..j] from L2 per thread: 39 GB/s at batch 65536);
//   * per chunk the tanh / knot search / cubic values are computed ONCE per (sample, feature) and shared through LDS by
//     the 8 threads of the sample;  the knots sit in LDS ("per-feature grid lookup in LDS", BASELINE.json north_star).
// fp32 throughout, VALU FMAs (4 of the num_basis products per (sample, feature, output) are non-zero; a dense MFMA
// form would multiply the structural zeros as well).
#include "common.h"

namespace {

constexpr int KS_MAX_KNOTS = 64;
constexpr int KS_MAX_LAYERS = 4;
constexpr int KS_G = 8;                    // threads per sample
constexpr int KS_MAXW = 64;                // widest hidden / output layer
constexpr int KS_SLAB_FLOATS = 8192;       // 32 KB weight slab per chunk

struct KanStackArgs {
  const float* x; int B;
  int nl;
  int dims[KS_MAX_LAYERS + 1];
  const float* W[KS_MAX_LAYERS];          // PREPARED (in, nb, out)
  const float* knots[KS_MAX_LAYERS];
  const float* lw[KS_MAX_LAYERS];         // PREPARED (in, out)
  const float* lb[KS_MAX_LAYERS];
  float* out[KS_MAX_LAYERS];              // (B, out) post-activation
  int nk[KS_MAX_LAYERS];
  int ic[KS_MAX_LAYERS];                  // input features per chunk
  int act[KS_MAX_LAYERS];
};

struct B4 { int j; float v[4]; };

// truncated cubic basis of kan.py:8-44 in closed form (SURVEY.md 8(a) addendum); knots in LDS
__device__ __forceinline__ B4 ks_basis(float xn, const float* knots, int nk, float inv_h0) {
  B4 r;
  const int nb = nk - 4;
  const float t0 = knots[0], tl = knots[nk - 1];
  const float xc = fminf(fmaxf(xn, t0), tl);                   // kan.py:16
  int j = (int)floorf((xc - t0) * inv_h0);
  j = j < 0 ? 0 : (j > nk - 1 ? nk - 1 : j);
  while (j > 0 && xc < knots[j]) --j;                          // exact search on the STORED knots (kan.py:24)
  while (j < nk - 1 && xc >= knots[j + 1]) ++j;
  if (j >= nb) { r.j = -1; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f; return r; }   // truncation: SURVEY.md 0.2
  const float tj = knots[j];
  const float u = (xc - tj) / (knots[j + 1] - tj);
  const float u2 = u * u, u3 = u2 * u, om = 1.f - u;
  r.j = j;
  r.v[0] = u3 * (1.f / 6.f);
  r.v[1] = (-3.f * u3 + 3.f * u2 + 3.f * u + 1.f) * (1.f / 6.f);
  r.v[2] = (3.f * u3 - 6.f * u2 + 4.f) * (1.f / 6.f);
  r.v[3] = om * om * om * (1.f / 6.f);
#pragma unroll
  for (int m = 0; m < 4; ++m)
    if (j - m < 0) r.v[m] = 0.f;                               // left edge loses terms
  return r;
}

// chunk prefetch into registers: slab (PF float4), linear weights (LPF float4), layer-0 inputs (XPF floats)
template <int TB>
__device__ __forceinline__ void ks_prefetch(const KanStackArgs& a, int l, int i0n, int IC, int in_f, int out_f, int nb, int b0, int tid,
                                            f32x4 (&pw)[KS_SLAB_FLOATS / 4 / (TB * KS_G)],
                                            f32x4 (&plw)[(16 * KS_MAXW / 4 + TB * KS_G - 1) / (TB * KS_G)], float (&px)[TB * 16 / (TB * KS_G)]) {
  constexpr int NT = TB * KS_G;
  constexpr int PF = KS_SLAB_FLOATS / 4 / NT, LPF = (16 * KS_MAXW / 4 + NT - 1) / NT, XPF = TB * 16 / NT;
  const int nin = min(IC, in_f - i0n);
  const int nw4 = nin * nb * out_f / 4, nlw4 = nin * out_f / 4;
  const f32x4* w4 = (const f32x4*)(a.W[l] + (size_t)i0n * nb * out_f);
  const f32x4* l4 = (const f32x4*)(a.lw[l] + (size_t)i0n * out_f);
#pragma unroll
  for (int u = 0; u < PF; ++u) { const int e = tid + u * NT; pw[u] = w4[e < nw4 ? e : (nw4 > 0 ? nw4 - 1 : 0)]; }
#pragma unroll
  for (int u = 0; u < LPF; ++u) { const int e = tid + u * NT; plw[u] = l4[e < nlw4 ? e : (nlw4 > 0 ? nlw4 - 1 : 0)]; }
  if (l == 0) {
#pragma unroll
    for (int u = 0; u < XPF; ++u) {
      const int e = tid + u * NT;
      const int sl = e / nin, il = e - sl * nin;
      const int bs_ = b0 + sl < a.B ? b0 + sl : a.B - 1;
      px[u] = (e < TB * nin) ? a.x[(size_t)bs_ * in_f + i0n + il] : 0.f;
    }
  }
}

// one layer of the stack; L is a compile-time layer index (a run-time index into the by-value argument struct would put
// the whole struct into scratch memory)
template <int TB, int L>
__device__ __forceinline__ void ks_layer(const KanStackArgs& a, float* s_knots, float* s_act, float* s_w, float* s_lw, float* s_bx, int* s_bj,
                                         float* s_bv, int tid, int bl, int g, int b0, int b) {
  constexpr int NT = TB * KS_G;
  constexpr int l = L;
    const int in_f = a.dims[l], out_f = a.dims[l + 1];
    const int nk = a.nk[l], nb = nk - 4, IC = a.ic[l];
    // LDS row stride (floats): rows of different basis index j would otherwise start on the same banks (64 floats = one
    // full bank period); 4 floats of padding stagger them.  Only for power-of-two widths >= 4 (cheap index arithmetic).
    const bool pad = out_f >= 4 && (out_f & (out_f - 1)) == 0;
    const int OS = pad ? out_f + 4 : out_f;
    const int sh4 = 31 - __clz(out_f >> 2 | 1);            // log2(out_f / 4)
    const int OPT = (out_f + KS_G - 1) / KS_G;             // outputs per thread (<= 8)
    const int o0 = g * OPT;
    const int no = max(0, min(OPT, out_f - o0));           // live outputs of this thread
    const float* src = l == 0 ? nullptr : s_act + ((l - 1) & 1) * TB * KS_MAXW;    // previous layer's outputs (LDS)
    __syncthreads();                                        // previous layer done with s_knots / slabs
    if (tid < nk) s_knots[tid] = a.knots[l][tid];
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (q < no) ? a.lb[l][o0 + q] : 0.f;
    __syncthreads();
    const float inv_h0 = 1.f / (s_knots[1] - s_knots[0]);

    // The slab, the linear weights and (layer 0) the inputs of chunk c+1 are loaded into registers while chunk c is
    // contracted, so the only exposed global-memory latency of a layer is its first chunk's.
    constexpr int PF = KS_SLAB_FLOATS / 4 / NT;            // float4 registers of slab prefetch per thread
    constexpr int LPF = (16 * KS_MAXW / 4 + NT - 1) / NT;
    constexpr int XPF = TB * 16 / NT;
    f32x4 pw[PF], plw[LPF];
    float px[XPF];
    // 16-byte copies need every chunk START (a multiple of IC features) and every chunk LENGTH (IC, and the short last chunk
    // in_f % IC) to be whole float4s, for the slab and for the linear weights; any other shape takes the scalar copy
    // (a final layer 6 -> 1 with 7 basis functions has 42 + 6 floats: the float4 copy would drop the last 2 + 2)
    const int tail = in_f % IC;
    const bool vec = ((nb * out_f * IC) & 3) == 0 && ((out_f * IC) & 3) == 0 &&
                     (tail == 0 || (((nb * out_f * tail) & 3) == 0 && ((out_f * tail) & 3) == 0));
    if (vec) ks_prefetch<TB>(a, l, 0, IC, in_f, out_f, nb, b0, tid, pw, plw, px);
    for (int i0 = 0; i0 < in_f; i0 += IC) {
      const int ni = min(IC, in_f - i0);
      const int nw = ni * nb * out_f, nlw = ni * out_f;
      // (1) chunk weights -> LDS
      if (vec) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int e = tid + u * NT;
          if (e < nw / 4) ((f32x4*)s_w)[pad ? e + (e >> sh4) : e] = pw[u];          // one float4 of padding per row
        }
#pragma unroll
        for (int u = 0; u < LPF; ++u) { const int e = tid + u * NT; if (e < nlw / 4) ((f32x4*)s_lw)[pad ? e + (e >> sh4) : e] = plw[u]; }
      } else {
        const float* wsrc = a.W[l] + (size_t)i0 * nb * out_f;
        const float* lsrc = a.lw[l] + (size_t)i0 * out_f;
        for (int e = tid; e < nw; e += NT) s_w[(e / out_f) * OS + e % out_f] = wsrc[e];
        for (int e = tid; e < nlw; e += NT) s_lw[(e / out_f) * OS + e % out_f] = lsrc[e];
      }
      // (2) tanh + grid lookup, once per (sample, feature)
#pragma unroll
      for (int u = 0; u < XPF; ++u) {
        const int e = tid + u * NT;
        if (e < TB * ni) {
          const int sl = e / ni, il = e - sl * ni;
          float xv = 0.f;
          B4 bs; bs.j = -1; bs.v[0] = bs.v[1] = bs.v[2] = bs.v[3] = 0.f;
          if (b0 + sl < a.B) {
            xv = l == 0 ? (vec ? px[u] : a.x[(size_t)(b0 + sl) * in_f + i0 + il]) : src[sl * KS_MAXW + i0 + il];
#if defined(KS_EXP) && KS_EXP == 1
            bs = ks_basis(xv * 0.3f, s_knots, nk, inv_h0);
#elif defined(KS_EXP) && KS_EXP == 3
            bs.j = 3; bs.v[0] = xv; bs.v[1] = bs.v[2] = bs.v[3] = 0.25f;
#else
            bs = ks_basis(tanhf(xv), s_knots, nk, inv_h0);
#endif
          }
          s_bx[sl * 16 + il] = xv;
          s_bj[sl * 16 + il] = bs.j;
          *(float4*)(s_bv + (sl * 16 + il) * 4) = make_float4(bs.v[0], bs.v[1], bs.v[2], bs.v[3]);
        }
      }
      __syncthreads();
      if (vec && i0 + IC < in_f) ks_prefetch<TB>(a, l, i0 + IC, IC, in_f, out_f, nb, b0, tid, pw, plw, px);   // in flight during the contraction
      // (3) contraction: out[o] += x * lw[o][i] + sum_m v[m] * W[i][o][j-m]
#if defined(KS_EXP) && KS_EXP == 2
      if (no > 100) {
#else
      if (no > 0) {
#endif
#pragma unroll 4
        for (int il = 0; il < ni; ++il) {             // unrolled: the LDS reads of four features are in flight together
          const float xv = s_bx[bl * 16 + il];
          const int j = s_bj[bl * 16 + il];
          const float4 v = *(const float4*)(s_bv + (bl * 16 + il) * 4);
          const float* lrow = s_lw + il * OS + o0;
          const float* wbase = s_w + (il * nb) * OS + o0;
          const int jc = j > 0 ? j : 0;
          const float* r0 = wbase + jc * OS;
          const float* r1 = wbase + (jc >= 1 ? jc - 1 : 0) * OS;
          const float* r2 = wbase + (jc >= 2 ? jc - 2 : 0) * OS;
          const float* r3 = wbase + (jc >= 3 ? jc - 3 : 0) * OS;
          const float v0 = j >= 0 ? v.x : 0.f, v1 = j >= 1 ? v.y : 0.f, v2 = j >= 2 ? v.z : 0.f, v3 = j >= 3 ? v.w : 0.f;
          if (OPT == 8) {                                  // 64 outputs: two 16-byte reads per row
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 lv = *(const float4*)(lrow + 4 * h);
              const float4 w0 = *(const float4*)(r0 + 4 * h), w1 = *(const float4*)(r1 + 4 * h);
              const float4 w2 = *(const float4*)(r2 + 4 * h), w3 = *(const float4*)(r3 + 4 * h);
              acc[4 * h + 0] = fmaf(v3, w3.x, fmaf(v2, w2.x, fmaf(v1, w1.x, fmaf(v0, w0.x, fmaf(xv, lv.x, acc[4 * h + 0])))));
              acc[4 * h + 1] = fmaf(v3, w3.y, fmaf(v2, w2.y, fmaf(v1, w1.y, fmaf(v0, w0.y, fmaf(xv, lv.y, acc[4 * h + 1])))));
              acc[4 * h + 2] = fmaf(v3, w3.z, fmaf(v2, w2.z, fmaf(v1, w1.z, fmaf(v0, w0.z, fmaf(xv, lv.z, acc[4 * h + 2])))));
              acc[4 * h + 3] = fmaf(v3, w3.w, fmaf(v2, w2.w, fmaf(v1, w1.w, fmaf(v0, w0.w, fmaf(xv, lv.w, acc[4 * h + 3])))));
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (q < no)
                acc[q] = fmaf(v3, r3[q], fmaf(v2, r2[q], fmaf(v1, r1[q], fmaf(v0, r0[q], fmaf(xv, lrow[q], acc[q])))));
          }
        }
      }
      __syncthreads();
    }
    // (4) activation, hand the layer output to the next layer (LDS) and to HBM
    float* dst = s_act + (l & 1) * TB * KS_MAXW + bl * KS_MAXW;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < no) {
        float z = acc[q];
        z = a.act[l] == ROVIT_ACT_RELU ? fmaxf(z, 0.f) : (a.act[l] == ROVIT_ACT_SIGMOID3 ? 3.f / (1.f + __expf(-z)) : z);
        dst[o0 + q] = z;
        if (b < a.B) a.out[l][(size_t)b * out_f + o0 + q] = z;
      }
    }
}

template <int TB>
__global__ __launch_bounds__(TB * KS_G) void kan_stack_fwd_kernel(const KanStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_knots = smem;                                   // [64]
  float* s_act = s_knots + KS_MAX_KNOTS;                   // [2][TB][64] layer outputs, ping-pong
  float* s_w = s_act + 2 * TB * KS_MAXW;                   // [IC][nb][out] spline slab of the chunk
  float* s_lw = s_w + KS_SLAB_FLOATS;                      // [IC][out] linear weights of the chunk
  // per (sample, chunk feature): x, interval j, 4 cubic values
  float* s_bx = s_lw + 16 * (KS_MAXW + 4);                 // [TB][ICmax = 16]
  int* s_bj = (int*)(s_bx + TB * 16);
  float* s_bv = (float*)(s_bj + TB * 16);                  // [TB][16][4]
  const int tid = threadIdx.x;
  const int bl = tid / KS_G, g = tid % KS_G;
  const int b0 = blockIdx.x * TB;
  const int b = b0 + bl;
  ks_layer<TB, 0>(a, s_knots, s_act, s_w, s_lw, s_bx, s_bj, s_bv, tid, bl, g, b0, b);
  if (a.nl > 1) ks_layer<TB, 1>(a, s_knots, s_act, s_w, s_lw, s_bx, s_bj, s_bv, tid, bl, g, b0, b);
  if (a.nl > 2) ks_layer<TB, 2>(a, s_knots, s_act, s_w, s_lw, s_bx, s_bj, s_bv, tid, bl, g, b0, b);
  if (a.nl > 3) ks_layer<TB, 3>(a, s_knots, s_act, s_w, s_lw, s_bx, s_bj, s_bv, tid, bl, g, b0, b);
}

// Prepared weight layouts of one KAN layer (re-run whenever the parameters change): spline_w (in, out, nb) -> spline_wt
// (in, nb, out); lin_w (out, in) -> lin_wt (in, out).
__global__ __launch_bounds__(256) void kan_prepare_kernel(const float* __restrict__ w, const float* __restrict__ lw, float* __restrict__ wt,
                                                          float* __restrict__ lwt, int in_f, int out_f, int nb) {
  const int n = in_f * out_f * nb;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {       // e indexes the OUTPUT (i, k, o): coalesced writes
    const int o = e % out_f, r = e / out_f, k = r % nb, i = r / nb;
    wt[e] = w[((size_t)i * out_f + o) * nb + k];
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < in_f * out_f; e += gridDim.x * 256) {
    const int o = e % out_f, i = e / out_f;
    lwt[e] = lw[(size_t)o * in_f + i];
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Dense matrix-core form of the same stack for LARGE batches (VERDICT r1 item 6 / DESIGN.md section 4):
//   layer(x)[b, o] = sum_j sum_s R[b, j, s] * Wd[j, s, o] + bias[o],   s = 0 .. 2H-1 "slots" of input feature j:
//   slots 0 .. nb-1 = the truncated cubic basis values B_s(tanh x[b, j]) (four of them non-zero), slot nb = the raw x[b, j]
//   (the layer's Linear term), the rest zero padding.  The contraction over (j, s) runs on v_mfma_f32_32x32x2_f32: fp32
//   operands and accumulation, i.e. the arithmetic class of the VALU kernels (only the summation order differs).  It
//   multiplies the structural zeros too (8/5 of the useful FMAs at G = 5, 36/5 at G = 32) but leaves the LDS-fed VALU loop
//   that bounds kan_stack_fwd_kernel.
// One wave = one workgroup = 32 samples for the whole stack (no barrier ever joins two waves):
//   M (rows of D) = 32 outputs of a tile, N (columns) = the 32 samples, K = 2 slots per instruction.
//   A operand (weights): lane l holds Wm[.., kk = l / 32][o = l % 32], read straight from the prepared global layout
//     Wm[j][q][t][kk][32] (rovit_kan_prepare_mfma): one coalesced 256-byte run per register, served by L1/L2 (all waves
//     walk the same weights in the same order);
//   B operand (slot values): lane l = (sample l % 32, kk = l / 32) reads its H values of slot parity kk from the dense
//     row R[b, j, :] that the wave built in LDS (rows are zeroed, then the <= 5 live values are written at their slot);
//   D: lane holds sample l % 32 and outputs 8 (r / 4) + 4 (l / 32) + r % 4 of the tile -> 16-byte stores.
// Layer outputs stay in LDS as the next layer's input and go to HBM once (backward / trajectory).
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KM_TS = 32;                  // samples per wave

// Basis values on a UNIFORM grid (the reference's knots are torch.linspace, kan.py:59; the caller checks it): the
// interval index comes from arithmetic instead of a search, its two knots are then read from the STORED fp32 knots (LDS),
// so u is the value ks_basis computes whenever the index agrees.  The spline is C2 across interior knots, so an interval
// chosen one ulp early or late moves the result by O(ulp^3); the ONE discontinuous decision -- x_c >= knots[nb] turns the
// whole basis off -- and the clamp use the stored knot values exactly (kcut, t0, tl).
__device__ __forceinline__ B4 km_basis(float xn, const float* knots, float t0, float tl, float kcut, float inv_h, int nb) {
  B4 r;
  const float xc = fminf(fmaxf(xn, t0), tl);
  int j = (int)floorf((xc - t0) * inv_h);
  j = j < 0 ? 0 : (j > nb - 1 ? nb - 1 : j);                  // a live sample sits left of knots[nb]
  const float tj = knots[j], tj1 = knots[j + 1];
  const float u = (xc - tj) / (tj1 - tj);
  const float u2 = u * u, u3 = u2 * u, om = 1.f - u;
  r.j = xc >= kcut ? -1 : j;
  r.v[0] = u3 * (1.f / 6.f);
  r.v[1] = (-3.f * u3 + 3.f * u2 + 3.f * u + 1.f) * (1.f / 6.f);
  r.v[2] = (3.f * u3 - 6.f * u2 + 4.f) * (1.f / 6.f);
  r.v[3] = om * om * om * (1.f / 6.f);
  return r;
}

struct KanMfmaArgs {
  const float* x; int B;
  int nl;
  int dims[KS_MAX_LAYERS + 1];
  const float* Wm[KS_MAX_LAYERS];         // PREPARED [in][H][NT][2][32]
  const float* knots[KS_MAX_LAYERS];
  const float* lb[KS_MAX_LAYERS];
  float* out[KS_MAX_LAYERS];
  int nk[KS_MAX_LAYERS];
  int act[KS_MAX_LAYERS];
  int as0, as1;                           // LDS row strides (floats) of the two activation buffers: outputs of even / odd layers
};

// H = half the padded slot count; FPL = features per lane and sub-chunk (a super-chunk is 8 features: lane (sample, h)
// owns features 4h .. 4h+3 of it and builds FPL of their rows per sub-chunk); NT = 32-output tiles of the layer.
template <int H, int FPL, int NS, int NT, int L>
__device__ __forceinline__ void km_layer(const KanMfmaArgs& a, float* s_knots, float* s_act, float* s_row, int lane, int b0) {
  constexpr int l = L;
  constexpr int S = 2 * H;
  constexpr int TS = KM_TS * NS;                             // samples of this wave: lane (smp, hk) serves samples smp + 32 s
  const int in_f = a.dims[l], out_f = a.dims[l + 1];
  const int nk = a.nk[l], nb = nk - 4;
  const int smp = lane & 31, hk = lane >> 5;
  __syncthreads();
  if (lane < nk) s_knots[lane] = a.knots[l][lane];
  __syncthreads();
  const float t0 = s_knots[0], tl = s_knots[nk - 1], kcut = s_knots[nb];
  const float inv_h = (float)(nk - 1) / (tl - t0);
  float* dummy = s_knots + KS_MAX_KNOTS + lane;              // sink of the scattered writes of absent basis terms
  // activation buffers: outputs of even layers in buffer 0 (stride as0), of odd layers in buffer 1 (stride as1)
  const int as_in = (l & 1) ? a.as0 : a.as1, as_out = (l & 1) ? a.as1 : a.as0;
  const float* src = l == 0 ? nullptr : s_act + ((l & 1) ? 0 : TS * a.as0) + smp * as_in;
  const float* xrow[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int bs = b0 + smp + 32 * s < a.B ? b0 + smp + 32 * s : a.B - 1;
    xrow[s] = l == 0 ? a.x + (size_t)bs * in_f : src + 32 * s * as_in;
  }
  const float* wl = a.Wm[l] + lane;
  f32x16 acc[NS][NT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;

  f32x4 xn[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) xn[s] = *(const f32x4*)(xrow[s] + 4 * hk);
  for (int i0 = 0; i0 < in_f; i0 += 8) {
    f32x4 xv4[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      xv4[s] = xn[s];
      if (i0 + 8 < in_f) xn[s] = *(const f32x4*)(xrow[s] + i0 + 8 + 4 * hk);
    }
#pragma unroll
    for (int sc = 0; sc < 4 / FPL; ++sc) {
      // ---- the sub-chunk's weights: issued first, they land while the rows are built
      float wr[2 * FPL][H * NT];
#pragma unroll
      for (int f = 0; f < 2 * FPL; ++f) {
        const int jin = i0 + 4 * (f & 1) + sc * FPL + (f >> 1);
        const float* wp = wl + (size_t)jin * H * NT * 64;
#pragma unroll
        for (int e = 0; e < H * NT; ++e) wr[f][e] = wp[e * 64];
      }
      // ---- build the dense slot rows of this sub-chunk: local feature f = 2 * u + hk  <->  input i0 + 4 * hk + sc * FPL + u
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int u = 0; u < FPL; ++u) {
          const float xv = xv4[s][sc * FPL + u];
          const B4 bq = km_basis(tanhf(xv), s_knots, t0, tl, kcut, inv_h, nb);
          float* row = s_row + ((2 * u + hk) * TS + 32 * s + smp) * S;
#pragma unroll
          for (int z = 0; z < S / 4; ++z) *(f32x4*)(row + 4 * z) = (f32x4){0.f, 0.f, 0.f, 0.f};
          // slot s lives at position (s & 1) * H + (s >> 1): the two slot parities are the two K rows of an MFMA step
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int sidx = bq.j - m;                       // < 0: left edge lost the term, or the sample is beyond the cutoff
            float* dst = sidx >= 0 ? row + ((sidx & 1) * H + (sidx >> 1)) : dummy;
            *dst = bq.v[m];
          }
          row[(nb & 1) * H + (nb >> 1)] = xv;
        }
      __syncthreads();
      // ---- contraction of the 2 * FPL features of the sub-chunk
#pragma unroll
      for (int f = 0; f < 2 * FPL; ++f) {
        float bv[NS][H];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float* rp = s_row + (f * TS + 32 * s + smp) * S + hk * H;
          if (H % 4 == 0) {
#pragma unroll
            for (int z = 0; z < H / 4; ++z) { const f32x4 v = *(const f32x4*)(rp + 4 * z); bv[s][4 * z] = v[0]; bv[s][4 * z + 1] = v[1]; bv[s][4 * z + 2] = v[2]; bv[s][4 * z + 3] = v[3]; }
          } else {
#pragma unroll
            for (int z = 0; z < H / 2; ++z) { const float2 v = *(const float2*)(rp + 2 * z); bv[s][2 * z] = v.x; bv[s][2 * z + 1] = v.y; }
          }
        }
#pragma unroll
        for (int q = 0; q < H; ++q)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s)
              acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[f][q * NT + t], bv[s][q], acc[s][t], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // ---- bias, activation; hand the outputs to the next layer (LDS) and to HBM
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float* dst = s_act + ((l & 1) ? TS * a.as0 : 0) + (32 * s + smp) * as_out;
    const int b = b0 + 32 * s + smp;
    const bool live = b < a.B;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int o = 32 * t + 8 * g + 4 * hk;
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[s][t][4 * g + e] + (o + e < out_f ? a.lb[l][o + e] : 0.f);
          v = a.act[l] == ROVIT_ACT_RELU ? fmaxf(v, 0.f) : (a.act[l] == ROVIT_ACT_SIGMOID3 ? 3.f / (1.f + __expf(-v)) : v);
          z[e] = v;
        }
        if (o < as_out) *(f32x4*)(dst + o) = (f32x4){z[0], z[1], z[2], z[3]};
        if (live) {
          if (o + 3 < out_f) *(f32x4*)(a.out[l] + (size_t)b * out_f + o) = (f32x4){z[0], z[1], z[2], z[3]};
          else
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (o + e < out_f) a.out[l][(size_t)b * out_f + o + e] = z[e];
        }
      }
  }
}

template <int H, int FPL, int NS, int L>
__device__ __forceinline__ void km_layer_nt(const KanMfmaArgs& a, float* s_knots, float* s_act, float* s_row, int lane, int b0) {
  if (a.dims[L + 1] > 32) km_layer<H, FPL, NS, 2, L>(a, s_knots, s_act, s_row, lane, b0);
  else km_layer<H, FPL, NS, 1, L>(a, s_knots, s_act, s_row, lane, b0);
}

// NS = 32-sample tiles per wave.  NS = 1: 18.7 KB (G = 5) / 19.7 KB (G = 32) of LDS for the default stack, eight workgroups per
// CU, all 2048 waves of a 65536-sample batch resident together.  NS = 2: every weight fragment fetched from L2 feeds two
// MFMAs (the kernel is bound by streaming the weights through L2 once per wave), 36 / 39 KB, four workgroups per CU.
template <int H, int FPL, int NS>
__global__ __launch_bounds__(64) void kan_stack_mfma_kernel(const KanMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float km_smem[];
  float* s_knots = km_smem;                                  // [64] knots + [64] write sink
  float* s_row = s_knots + 2 * KS_MAX_KNOTS;                 // [2 FPL][32 NS][2H]
  float* s_act = s_row + 2 * FPL * KM_TS * NS * 2 * H;       // [32 NS][as0] + [32 NS][as1]
  const int lane = threadIdx.x;
  const int b0 = blockIdx.x * KM_TS * NS;
  km_layer_nt<H, FPL, NS, 0>(a, s_knots, s_act, s_row, lane, b0);
  if (a.nl > 1) km_layer_nt<H, FPL, NS, 1>(a, s_knots, s_act, s_row, lane, b0);
  if (a.nl > 2) km_layer_nt<H, FPL, NS, 2>(a, s_knots, s_act, s_row, lane, b0);
  if (a.nl > 3) km_layer_nt<H, FPL, NS, 3>(a, s_knots, s_act, s_row, lane, b0);
}

// half the padded slot count of a layer with nb basis functions (+1 slot for the Linear term), rounded up to even
__host__ __device__ inline int km_half_slots(int nb) { const int h = (nb + 2) / 2; return (h + 1) & ~1; }

// Wm[j][q][t][kk][o'] = Wd[j][s = 2q + kk][o = 32 t + o'],  Wd[j][s][o] = spline_w[j][o][s] (s < nb), lin_w[o][j] (s == nb), else 0
__global__ __launch_bounds__(256) void kan_prepare_mfma_kernel(const float* __restrict__ w, const float* __restrict__ lw, float* __restrict__ wm,
                                                               int in_f, int out_f, int nb, int H, int NT) {
  const size_t n = (size_t)in_f * H * NT * 64;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int op = (int)(e & 31), kk = (int)(e >> 5) & 1;
    size_t r = e >> 6;
    const int t = (int)(r % NT); r /= NT;
    const int q = (int)(r % H);
    const int j = (int)(r / H);
    const int s = 2 * q + kk, o = 32 * t + op;
    float v = 0.f;
    if (o < out_f) v = s < nb ? w[((size_t)j * out_f + o) * nb + s] : (s == nb ? lw[(size_t)o * in_f + j] : 0.f);
    wm[e] = v;
  }
}

}  // namespace

extern "C" int rovit_kan_prepare(const float* spline_w, const float* lin_w, float* spline_wt, float* lin_wt, int in_f, int out_f,
                                 int n_basis, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(spline_w && lin_w && spline_wt && lin_wt, ROVIT_ERR_NULL, "kan_prepare: null pointer");
  ROVIT_CHECK_ARG(in_f > 0 && out_f > 0 && n_basis > 0, ROVIT_ERR_SHAPE, "kan_prepare: bad shape");
  ROVIT_CHECK_ARG(rovit_aligned16(spline_wt) && rovit_aligned16(lin_wt), ROVIT_ERR_ALIGN, "kan_prepare: outputs must be 16-byte aligned");
  const int n = in_f * out_f * n_basis;
  int blocks = (n + 255) / 256;
  blocks = blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL(kan_prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, spline_w, lin_w, spline_wt, lin_wt, in_f, out_f, n_basis);
  ROVIT_CHECK_LAUNCH("kan_prepare_kernel");
  return ROVIT_OK;
}

// spline_wt / lin_wt: the PREPARED layouts of rovit_kan_prepare.
// spline_wt / knots / lin_wt / lin_b / outs: HOST arrays of n_layers device pointers; dims: n_layers + 1 widths;
// acts: activation after each layer (ROVIT_ACT_*).  outs[l] (batch, dims[l+1]) receives the post-activation output of
// layer l (the last one is the module output; the others are what backward / get_activation_trajectory need).
extern "C" int rovit_kan_stack_fwd(const float* x, const float* const* spline_w, const float* const* knots, const float* const* lin_w,
                                   const float* const* lin_b, float* const* outs, int batch, const int* dims, const int* n_knots,
                                   const int* acts, int n_layers, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && spline_w && knots && lin_w && lin_b && outs && dims && n_knots && acts, ROVIT_ERR_NULL, "kan_stack_fwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && n_layers >= 1 && n_layers <= KS_MAX_LAYERS, ROVIT_ERR_SHAPE, "kan_stack_fwd: 1..%d layers", KS_MAX_LAYERS);
  KanStackArgs a{};
  a.x = x; a.B = batch; a.nl = n_layers;
  for (int l = 0; l <= n_layers; ++l) a.dims[l] = dims[l];
  for (int l = 0; l < n_layers; ++l) {
    ROVIT_CHECK_ARG(spline_w[l] && knots[l] && lin_w[l] && lin_b[l] && outs[l], ROVIT_ERR_NULL, "kan_stack_fwd: null pointer in layer %d", l);
    ROVIT_CHECK_ARG(rovit_aligned16(spline_w[l]) && rovit_aligned16(lin_w[l]), ROVIT_ERR_ALIGN, "kan_stack_fwd: prepared weights must be 16-byte aligned");
    ROVIT_CHECK_ARG(dims[l] > 0 && dims[l + 1] > 0 && dims[l + 1] <= KS_MAXW && (l == 0 || dims[l] <= KS_MAXW), ROVIT_ERR_SHAPE,
                    "kan_stack_fwd: layer %d is %d -> %d; widths after the input must be <= %d", l, dims[l], dims[l + 1], KS_MAXW);
    ROVIT_CHECK_ARG(dims[l + 1] % 4 == 0 || dims[l + 1] < KS_G, ROVIT_ERR_SHAPE, "kan_stack_fwd: output width %d must be a multiple of 4 (or < 8)",
                    dims[l + 1]);
    ROVIT_CHECK_ARG(n_knots[l] >= 8 && n_knots[l] <= KS_MAX_KNOTS, ROVIT_ERR_SHAPE, "kan_stack_fwd: degree-3 layer needs 8..%d knots", KS_MAX_KNOTS);
    a.W[l] = spline_w[l]; a.knots[l] = knots[l]; a.lw[l] = lin_w[l]; a.lb[l] = lin_b[l]; a.out[l] = outs[l];
    a.nk[l] = n_knots[l]; a.act[l] = acts[l];
    const int per_feature = (n_knots[l] - 4) * (dims[l + 1] + 4);
    int ic = KS_SLAB_FLOATS / per_feature;
    ic = ic > 16 ? 16 : ic;
    ROVIT_CHECK_ARG(ic >= 1, ROVIT_ERR_SHAPE, "kan_stack_fwd: layer %d: one feature's slab does not fit the LDS tile", l);
    a.ic[l] = ic;
  }
  // small batches: 16 samples per workgroup so that more CUs take part; large batches: 32 (65 KB of LDS: two
  // workgroups = 8 waves per CU, one in its staging phase while the other contracts)
  const bool small = batch <= 4096;
  const int tb = small ? 16 : 32;
  const size_t lds = (KS_MAX_KNOTS + 2 * (size_t)tb * KS_MAXW + KS_SLAB_FLOATS + 16 * (KS_MAXW + 4) + (size_t)tb * 16 * 6) * sizeof(float);
  if (small) {
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)kan_stack_fwd_kernel<16>, lds), ROVIT_ERR_LAUNCH, "kan_stack_fwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(kan_stack_fwd_kernel<16>, dim3((batch + tb - 1) / tb), dim3(tb * KS_G), lds, (hipStream_t)stream, a);
  } else {
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)kan_stack_fwd_kernel<32>, lds), ROVIT_ERR_LAUNCH, "kan_stack_fwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(kan_stack_fwd_kernel<32>, dim3((batch + tb - 1) / tb), dim3(tb * KS_G), lds, (hipStream_t)stream, a);
  }
  ROVIT_CHECK_LAUNCH("kan_stack_fwd_kernel");
  return ROVIT_OK;
}

// ---- matrix-core form (large batches) ------------------------------------------------------------------------------
// floats of the prepared MFMA weight layout of one layer, or 0 when the layer shape is not supported by the kernel
extern "C" size_t rovit_kan_mfma_prepared_floats(int in_f, int out_f, int n_basis) {
  if (in_f <= 0 || out_f <= 0 || out_f > KS_MAXW || n_basis < 4 || in_f % 8 != 0) return 0;
  const int H = km_half_slots(n_basis);
  if (H != 4 && H != 18) return 0;                    // instantiated: G = 5 (7 basis -> 8 slots) and G = 32 (34 -> 36)
  const int NT = out_f > 32 ? 2 : 1;
  return (size_t)in_f * H * NT * 64;
}

extern "C" int rovit_kan_prepare_mfma(const float* spline_w, const float* lin_w, float* wm, int in_f, int out_f, int n_basis,
                                      rovit_stream_t stream) {
  ROVIT_CHECK_ARG(spline_w && lin_w && wm, ROVIT_ERR_NULL, "kan_prepare_mfma: null pointer");
  const size_t n = rovit_kan_mfma_prepared_floats(in_f, out_f, n_basis);
  ROVIT_CHECK_ARG(n > 0, ROVIT_ERR_SHAPE, "kan_prepare_mfma: unsupported layer %d -> %d with %d basis functions", in_f, out_f, n_basis);
  ROVIT_CHECK_ARG(rovit_aligned16(wm), ROVIT_ERR_ALIGN, "kan_prepare_mfma: output must be 16-byte aligned");
  int blocks = (int)((n + 255) / 256);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(kan_prepare_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, spline_w, lin_w, wm, in_f, out_f, n_basis,
                     km_half_slots(n_basis), out_f > 32 ? 2 : 1);
  ROVIT_CHECK_LAUNCH("kan_prepare_mfma_kernel");
  return ROVIT_OK;
}

// wm: HOST array of n_layers device pointers to the layouts of rovit_kan_prepare_mfma; other arguments as rovit_kan_stack_fwd.
// All layers must share the knot count (one kernel instantiation per slot count).
extern "C" int rovit_kan_stack_fwd_mfma(const float* x, const float* const* wm, const float* const* knots, const float* const* lin_b,
                                        float* const* outs, int batch, const int* dims, const int* n_knots, const int* acts, int n_layers,
                                        rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && wm && knots && lin_b && outs && dims && n_knots && acts, ROVIT_ERR_NULL, "kan_stack_fwd_mfma: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && n_layers >= 1 && n_layers <= KS_MAX_LAYERS, ROVIT_ERR_SHAPE, "kan_stack_fwd_mfma: 1..%d layers", KS_MAX_LAYERS);
  ROVIT_CHECK_ARG(rovit_aligned16(x), ROVIT_ERR_ALIGN, "kan_stack_fwd_mfma: x must be 16-byte aligned");
  KanMfmaArgs a{};
  a.x = x; a.B = batch; a.nl = n_layers;
  for (int l = 0; l <= n_layers; ++l) a.dims[l] = dims[l];
  const int H = km_half_slots(n_knots[0] - 4);
  for (int l = 0; l < n_layers; ++l) {
    ROVIT_CHECK_ARG(wm[l] && knots[l] && lin_b[l] && outs[l], ROVIT_ERR_NULL, "kan_stack_fwd_mfma: null pointer in layer %d", l);
    ROVIT_CHECK_ARG(n_knots[l] == n_knots[0] && n_knots[l] <= KS_MAX_KNOTS, ROVIT_ERR_SHAPE, "kan_stack_fwd_mfma: layers must share the knot count");
    ROVIT_CHECK_ARG(rovit_kan_mfma_prepared_floats(dims[l], dims[l + 1], n_knots[l] - 4) > 0 && (l == 0 || dims[l] <= KS_MAXW), ROVIT_ERR_SHAPE,
                    "kan_stack_fwd_mfma: unsupported layer %d: %d -> %d, %d knots", l, dims[l], dims[l + 1], n_knots[l]);
    ROVIT_CHECK_ARG(rovit_aligned16(wm[l]) && rovit_aligned16(outs[l]) && (dims[l + 1] % 4 == 0 || dims[l + 1] < 4), ROVIT_ERR_ALIGN,
                    "kan_stack_fwd_mfma: layer %d: buffers must be 16-byte aligned, widths a multiple of 4 (or < 4)", l);
    a.Wm[l] = wm[l]; a.knots[l] = knots[l]; a.lb[l] = lin_b[l]; a.out[l] = outs[l]; a.nk[l] = n_knots[l]; a.act[l] = acts[l];
  }
  // activation rows hold whole 4-float groups of the widest layer they serve (a layer's 32-output tiles are cut at that width)
  for (int l = 0; l < n_layers; ++l) {
    const int w = (dims[l + 1] + 3) & ~3;
    if (l & 1) a.as1 = w > a.as1 ? w : a.as1; else a.as0 = w > a.as0 ? w : a.as0;
  }
  if (a.as1 == 0) a.as1 = 4;
  const int fpl = H == 4 ? 4 : 1;
  // two sample tiles per wave halve the weight traffic through L2 but leave one wave per SIMD: pays when the MFMA phase
  // dominates and the batch still gives every SIMD a wave (G = 32 at batch 65536: 696 -> 618 us; G = 5: 186 -> 210 us, not used)
  const int ns_env = ROVIT_KNOB(ROVIT_KNOB_KAN_MFMA_NS, 0);
  ROVIT_CHECK_ARG(ns_env >= 0 && ns_env <= 2, ROVIT_ERR_SHAPE, "kan_stack_fwd_mfma: ROVIT_KAN_MFMA_NS must be 1 or 2 (got %d)", ns_env);
  const int ns = ns_env ? ns_env : ((H == 18 && batch >= 49152) ? 2 : 1);
  const int grid = (batch + KM_TS * ns - 1) / (KM_TS * ns);
  const size_t lds = (2 * KS_MAX_KNOTS + (size_t)2 * fpl * KM_TS * ns * 2 * H + (size_t)KM_TS * ns * (a.as0 + a.as1)) * sizeof(float);
  if (H == 4 && ns == 1) hipLaunchKernelGGL((kan_stack_mfma_kernel<4, 4, 1>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
  else if (H == 4) hipLaunchKernelGGL((kan_stack_mfma_kernel<4, 4, 2>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
  else if (ns == 1) hipLaunchKernelGGL((kan_stack_mfma_kernel<18, 1, 1>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((kan_stack_mfma_kernel<18, 1, 2>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("kan_stack_mfma_kernel");
  return ROVIT_OK;
}
