// bf16 MFMA GEMMs for the DeiT-Tiny token matrix (M = B*197 rows, K/N in {192, 576, 768}).
//
//   gemm_nt   C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)       forward linears and dgrads
//   wgrad     G[N,K] = dY[M,N]^T * A[M,K], colsum(dY)               weight gradients (split over M, slabs)
//
// Kernels, in the order they appear (all persistent and weight-stationary except the first):
//   gemm_nt_kernel      128x192 / 128x96 tiles, operands through LDS: generic fallback (rovit_set_gemm_tile)
//   gemm_ws_kernel      W fragments in registers, A tiles register-staged into a double-buffered LDS tile;
//                       K = 768 splits K over wave pairs.  Used for the row-wise LayerNorm epilogues at K = 192 / 768
//                       and the patch embedding
//   gemm_ws_dma_kernel  K = 192: A tiles by LDS-DMA into a 3-slot ring (qkv fwd, fc1 fwd + GELU, proj dgrad) and, with
//                       a second ring for the elementwise factor, fc2 dgrad x gelu'
//   gemm_kdma_kernel    K = 576: 12 waves x 16 columns over the whole K, LDS-DMA ring, row-wise epilogue
//                       (qkv dgrad + LayerNorm backward)
//   wgrad_kernel, wgrad_reduce_batch_kernel (slab sums + LayerNorm-affine un-fold), wgrad_affine_finalize_kernel
//
// Reference arithmetic being restated: timm VisionTransformer's Linear layers (qkv/proj/fc1/fc2, patch-embed
// conv as an im2col GEMM) reached through /root/reference/models/backbone.py:12-25, and their autograd
// backward (training/trainer.py:119,136).
//
// MFMA: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  The accumulator is computed TRANSPOSED (weights as the
// MFMA "A" operand) so that each lane ends up with 4 consecutive output columns of one row: 8-byte bf16 /
// 16-byte fp32 epilogue accesses.  Workgroup ids are remapped so that all column tiles of one row panel (and
// all output tiles of one M-split in wgrad) run on the same XCD and share that XCD's L2.
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifdef ROVIT_DEV
#define GEMM_DBG(g, bit) ((g).dbg & (bit))
#define GEMM_SET_DBG(g, v) (g).dbg = (v)
#else
#define GEMM_DBG(g, bit) (false)    // the product kernels have no skip-work path
#define GEMM_SET_DBG(g, v) ((void)0)
#endif

namespace {

inline bool set_max_lds(const void* fn, size_t bytes) { return rovit_set_max_lds(fn, bytes); }

constexpr int BK = 64;
constexpr int LDS_STRIDE = BK + 16;      // bf16 elements; 160-byte rows: conflict-free ds_read_b128 fragments

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_MUL = 3, EPI_PATCH = 4, EPI_RESID_LN = 5, EPI_LNBWD = 6,
       EPI_PATCH_IMG = 7 };   // internal: EPI_PATCH with the A operand gathered from the fp32 NCHW images (rovit_patch_embed_fwd)

struct GemmArgs {
  const bf16* A; int lda;
  const bf16* W; int ldw;
  int M, N, K;
  const float* bias;
  bf16* out; int ldo;
  bf16* out2;
  float* xres; int ldx;
  const bf16* mul; int ldm;
  const float* pos; int tokens;
  int n_tiles;
  float* rstd_out; float eps;   // EPI_RESID_LN: statistics of the LayerNorm fused behind the residual add
  const bf16* xprev;  // EPI_LNBWD, round 4: the incoming residual gradient as bf16 rows (ld 192); then xres is neither read nor written
  const float* img;   // EPI_PATCH_IMG: images (B,3,224,224) fp32; row m = (image, patch), column = c*256 + kh*16 + kw
#ifdef ROVIT_DEV
  int dbg;            // developer library only (ROVIT_KNOB_GEMM_DBG): bit 0 skip epilogue stores, 1 skip MFMAs, 2 skip DMA, 3 skip GELU
#endif
};

// gelu_and_grad / gelu_grad: common.h (shared with mlp_fused.hip)

template <int EPI>
__device__ __forceinline__ void epilogue4(const GemmArgs& g, int m, int n, f32x4 v) {
  if (g.bias) {
    const float4 b = *(const float4*)(g.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (EPI == EPI_BF16) {
    *(bf16x4*)(g.out + (size_t)m * g.ldo + n) = pack4(v);
  } else if (EPI == EPI_GELU) {
    f32x4 a, d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float ga, gd;
      gelu_and_grad(v[r], ga, gd);                                         // exact-erf GELU (timm default)
      a[r] = ga; d[r] = gd;
    }
    *(bf16x4*)(g.out + (size_t)m * g.ldo + n) = pack4(a);
    if (g.out2) *(bf16x4*)(g.out2 + (size_t)m * g.ldo + n) = pack4(d);
  } else if (EPI == EPI_RESID) {
    float4* p = (float4*)(g.xres + (size_t)m * g.ldx + n);
    float4 x = *p;
    x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
    *p = x;
  } else if (EPI == EPI_MUL) {
    const bf16x4 q = *(const bf16x4*)(g.mul + (size_t)m * g.ldm + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= (float)q[r];
    *(bf16x4*)(g.out + (size_t)m * g.ldo + n) = pack4(v);
  } else if (EPI == EPI_PATCH || EPI == EPI_PATCH_IMG) {
    const int np = g.tokens - 1;
    const int b = m / np, p = m - b * np;
    const float4 pe = *(const float4*)(g.pos + (size_t)(p + 1) * g.N + n);
    float4 x = make_float4(v[0] + pe.x, v[1] + pe.y, v[2] + pe.z, v[3] + pe.w);
    *(float4*)(g.xres + ((size_t)b * g.tokens + 1 + p) * g.ldx + n) = x;
  }
}

// BM x BN output tile, WM x WN waves, K looped in steps of 64 through a double-buffered LDS ring filled by
// register staging (global_load_dwordx4 -> ds_write_b128 after the compute of the previous tile).
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const GemmArgs g) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int CPR = BK / 8;                       // 16-byte chunks per tile row
  constexpr int CA = BM * CPR / NT, CB = BN * CPR / NT;
  static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* As = lds;                                   // [2][BM][LDS_STRIDE]
  bf16* Bs = lds + 2 * BM * LDS_STRIDE;             // [2][BN][LDS_STRIDE]

  // XCD-aware tile id: ids b, b+8, ... share an XCD; give one XCD all column tiles of a row panel.
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / g.n_tiles, nt = bid - mt * g.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= g.M) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l15 = lane & 15, lg = lane >> 4;

  const bf16* a_src[CA];
  const bf16* b_src[CB];
  int a_dst[CA], b_dst[CB];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + i * NT, row = c / CPR, kc = c - row * CPR;
    int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
    a_src[i] = g.A + (size_t)gr * g.lda + kc * 8;
    a_dst[i] = row * LDS_STRIDE + kc * 8;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int c = tid + i * NT, row = c / CPR, kc = c - row * CPR;
    b_src[i] = g.W + (size_t)(n0 + row) * g.ldw + kc * 8;
    b_dst[i] = row * LDS_STRIDE + kc * 8;
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 ra[CA], rb[CB];
  const int nk = g.K / BK;
#pragma unroll
  for (int i = 0; i < CA; ++i) ra[i] = *(const bf16x8*)(a_src[i]);
#pragma unroll
  for (int i = 0; i < CB; ++i) rb[i] = *(const bf16x8*)(b_src[i]);
#pragma unroll
  for (int i = 0; i < CA; ++i) *(bf16x8*)(As + a_dst[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < CB; ++i) *(bf16x8*)(Bs + b_dst[i]) = rb[i];
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < CA; ++i) ra[i] = *(const bf16x8*)(a_src[i] + (kt + 1) * BK);
#pragma unroll
      for (int i = 0; i < CB; ++i) rb[i] = *(const bf16x8*)(b_src[i] + (kt + 1) * BK);
    }
    const bf16* Ac = As + cur * BM * LDS_STRIDE + (wm * WTM + l15) * LDS_STRIDE + lg * 8;
    const bf16* Bc = Bs + cur * BN * LDS_STRIDE + (wn * WTN + l15) * LDS_STRIDE + lg * 8;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(Ac + i * 16 * LDS_STRIDE + ks * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8*)(Bc + j * 16 * LDS_STRIDE + ks * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);   // D[n][m]
    }
    if (kt + 1 < nk) {
      bf16* An = As + (cur ^ 1) * BM * LDS_STRIDE;
      bf16* Bn = Bs + (cur ^ 1) * BN * LDS_STRIDE;
#pragma unroll
      for (int i = 0; i < CA; ++i) *(bf16x8*)(An + a_dst[i]) = ra[i];
#pragma unroll
      for (int i = 0; i < CB; ++i) *(bf16x8*)(Bn + b_dst[i]) = rb[i];
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + l15;
    if (m < g.M) {
#pragma unroll
      for (int j = 0; j < TN; ++j) epilogue4<EPI>(g, m, n0 + wn * WTN + j * 16 + lg * 4, acc[i][j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Weight-stationary persistent GEMM for the shapes this model actually has (N a multiple of 192, K in
// {192, 576, 768}).  A workgroup owns one 192-column chunk of W: wave (wn, wk) keeps W[48 wn .. +48][its K slice]
// as MFMA fragments in REGISTERS for its whole life and walks a contiguous range of BM-row tiles of A.  Only A
// goes through LDS (double-buffered, next tile's global loads in flight during the current tile's MFMAs), so
// LDS traffic is one fragment read per 3 MFMAs and there is one barrier per tile, not per K step.
// WK == 2 splits K over two waves per column slice; the pair exchanges partial accumulators through LDS and
// each finishes half of the tile's rows.
// ------------------------------------------------------------------------------------------------------
// im2col on the fly (PatchEmbed conv k16 s16 as a GEMM, timm VisionTransformer via /root/reference/models/backbone.py:12-25): the eight
// consecutive K columns col .. col+7 of patch row m are eight consecutive floats of one image row
struct F8 { f32x4 lo, hi; };
__device__ __forceinline__ const float* patch_src(const float* img, int m, int col) {
  const int b = m / 196, p = m - b * 196;
  const int ph = p / 14, pw = p - ph * 14;
  const int c = col >> 8, kh = (col >> 4) & 15, kw = col & 15;
  return img + (((size_t)b * 3 + c) * 224 + ph * 16 + kh) * 224 + pw * 16 + kw;
}
__device__ __forceinline__ F8 load_f8(const float* p) { F8 r; r.lo = *(const f32x4*)p; r.hi = *(const f32x4*)(p + 4); return r; }

template <int KS, int WK, int BM, int EPI>
__global__ __launch_bounds__(256 * WK, 2) void gemm_ws_kernel(const GemmArgs g, int tiles_per_wg, int n_tiles_m) {
  constexpr int K = KS * 32 * WK;
  constexpr int NT = 256 * WK;
  constexpr int STR = K + 16;                  // LDS row stride: conflict-free ds_read_b128 for K = 192/576/768
  constexpr int TM = BM / 16;
  constexpr int CPR = K / 8;
  constexpr int NCH = BM * CPR;
  constexpr int CH = (NCH + NT - 1) / NT;
  constexpr int TOWN = TM / WK;                // row tiles each wave finishes
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* As = lds;                              // [2][BM][STR]
  f32x4* xch = (f32x4*)(lds + 2 * BM * STR);   // WK==2: [2][8 waves][TOWN*3][64]
  constexpr int CSTR = 192 + 8;                // staged output tile [BM][CSTR] bf16
  bf16* Cs = lds + 2 * BM * STR + (WK == 2 ? 2 * 8 * TOWN * 3 * 64 * 8 : 0);

  // ids that share an XCD get the same row range and different column chunks
  const int nchunks = g.n_tiles;
  const int bid = blockIdx.x;
  const int chunk = (bid >> 3) % nchunks;
  const int p = (bid / (8 * nchunks)) * 8 + (bid & 7);
  const int tile0 = p * tiles_per_wg;
  int ntile = n_tiles_m - tile0;
  ntile = ntile > tiles_per_wg ? tiles_per_wg : ntile;
  if (ntile <= 0) return;
  const int n0 = chunk * 192;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 3, wk = wave >> 2;
  const int l15 = lane & 15, lg = lane >> 4;

  // stationary W fragments: rows (output columns) n0 + 48 wn + 16 j + l15, k = wk*KS*32 + 32 ks + 8 lg
  bf16x8 wf[3][KS];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[j][ks] = *(const bf16x8*)(g.W + (size_t)(n0 + 48 * wn + 16 * j + l15) * g.ldw + wk * KS * 32 + ks * 32 + lg * 8);

  int c_row[CH], c_col[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + i * NT;
    if (EPI == EPI_PATCH_IMG) {
      // image gather: the tile ROW runs fastest over the lanes, so that a wave reads 32 neighbouring patches x the two
      // 32-byte halves of one 16-pixel image-row segment = contiguous 2 KB runs (row-major chunk order reads 32 bytes
      // out of every 896: 94 us instead of 41 + 42 for the im2col pass and its GEMM)
      c_row[i] = c % BM; c_col[i] = (c / BM) * 8;
    } else {
      c_row[i] = c / CPR; c_col[i] = (c - c_row[i] * CPR) * 8;
    }
  }
  // Two register sets keep the global loads of tiles t+1 and t+2 in flight while tile t is computed: with a
  // single set the next load could only be issued after the previous one had landed (one tile per workgroup in
  // flight = ~25 GB/s per CU, the measured ceiling of the first version of this kernel).
  constexpr bool IMG = (EPI == EPI_PATCH_IMG);     // A gathered from fp32 images: the raw floats wait in registers, packed at the LDS store
  using RV = typename std::conditional<IMG, F8, bf16x8>::type;
  RV rvA[CH], rvB[CH];
  auto gload = [&](int tile, RV* rv) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (NCH % NT == 0 || tid + i * NT < NCH) {
        int gr = tile * BM + c_row[i];
        gr = gr < g.M ? gr : g.M - 1;
        if constexpr (IMG) rv[i] = load_f8(patch_src(g.img, gr, c_col[i]));
        else rv[i] = *(const bf16x8*)(g.A + (size_t)gr * g.lda + c_col[i]);
      }
    }
  };
  auto lstore = [&](int buf, const RV* rv) {
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (NCH % NT == 0 || tid + i * NT < NCH) {
        if constexpr (IMG) *(bf16x8*)(As + (buf * BM + c_row[i]) * STR + c_col[i]) = pack8(rv[i].lo, rv[i].hi);
        else *(bf16x8*)(As + (buf * BM + c_row[i]) * STR + c_col[i]) = rv[i];
      }
  };

  gload(tile0, rvA);
  lstore(0, rvA);
  if (ntile > 1) gload(tile0 + 1, rvA);
  // A second register set (two tiles in flight per workgroup) was measured SLOWER here (spills at the 256-VGPR
  // budget: qkv 26.4 -> 30.7 us), so it stays off; the code path is kept for shapes with a smaller W slice.
  constexpr bool TWO_SETS = false;
  if (TWO_SETS && ntile > 2) gload(tile0 + 2, rvB);
  barrier_lds();

  auto tile_body = [&](int t, int cur) {
    f32x4 acc[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16* Ac = As + (cur * BM + l15) * STR + wk * KS * 32 + lg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(Ac + i * 16 * STR + ks * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(wf[j][ks], af[i], acc[i][j]);     // D[n][m]
    }
    const int mbase = GEMM_DBG(g, 1) ? g.M : (tile0 + t) * BM;
    constexpr bool STAGED = (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_MUL || EPI == EPI_RESID_LN || EPI == EPI_LNBWD);
    constexpr bool ROWWISE = (EPI == EPI_RESID_LN || EPI == EPI_LNBWD);
    if (WK == 2) {
      // partner = same wn, other wk.  Wave wk finishes row tiles i with (i % WK) == wk; it ships the others.
      f32x4* xo = xch + ((size_t)(cur * 8 + (wave ^ 4)) * TOWN * 3) * 64 + lane;    // partner's inbox
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((i % WK) != wk) {
#pragma unroll
          for (int j = 0; j < 3; ++j) xo[((i / WK) * 3 + j) * 64] = acc[i][j];
        }
      }
      barrier_lds();
      const f32x4* xi = xch + ((size_t)(cur * 8 + wave) * TOWN * 3) * 64 + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((i % WK) == wk) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const f32x4 o = xi[((i / WK) * 3 + j) * 64];
            acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
          }
        }
      }
    }
    if (!STAGED) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((i % WK) == wk) {
          const int m = mbase + i * 16 + l15;
          if (m < g.M) {
#pragma unroll
            for (int j = 0; j < 3; ++j) epilogue4<EPI>(g, m, n0 + 48 * wn + 16 * j + 4 * lg, acc[i][j]);
          }
        }
      }
      if (WK == 1) barrier_lds();
    } else {
      // bf16 outputs: stage the (acc + bias) tile in LDS, then write whole 384-byte rows with 16-byte lanes
      // (the accumulator layout alone would give 16 rows x 32 bytes per store instruction).
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if ((i % WK) == wk) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            f32x4 v = acc[i][j];
            const int nl = 48 * wn + 16 * j + 4 * lg;
            if (g.bias) {
              const float4 bb = *(const float4*)(g.bias + n0 + nl);
              v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
            }
            *(bf16x4*)(Cs + (i * 16 + l15) * CSTR + nl) = pack4(v);
          }
        }
      }
      barrier_lds();
      if (ROWWISE) {
        // Row-wise epilogues (N = 192 = one full LayerNorm row per tile row): 16 lanes per row, lane c holds
        // elements {64 i + 4 c .. +3}; statistics by xor-shuffles inside the 16-lane group.
        constexpr int RPP = NT / 16;
        const int c = tid & 15;
#pragma unroll
        for (int pass = 0; pass < BM / RPP; ++pass) {
          const int row = pass * RPP + (tid >> 4);
          const int m = mbase + row;
          if (m < g.M) {
            float v[12];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const bf16x4 t = *(const bf16x4*)(Cs + row * CSTR + 64 * i + 4 * c);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)t[e];
            }
            float4* xp = (float4*)(g.xres + (size_t)m * g.ldx);
            if (EPI == EPI_RESID_LN) {
              // X += bf16(branch output); then (optionally) the next LayerNorm: xhat (bf16) and rstd.
              // All arithmetic first, all stores last.  (With the X stores in the middle, one build of this pass
              // -- three back-to-back 16-byte stores followed at once by packed adds that recycled the stores'
              // address registers -- produced a wrong row sum in lanes 48-63 of a wave in ~15 % of launches at
              // M = 50432, on two different MI355X; see DESIGN.md "observed hazard".  tools/stress_ln.py and
              // tests/test_gpu_stress.py screen for it.)
              float4 xs[3];
              float sum = 0.f;
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                float4 x = xp[16 * i + c];
                x.x += v[4 * i]; x.y += v[4 * i + 1]; x.z += v[4 * i + 2]; x.w += v[4 * i + 3];
                xs[i] = x;
                v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
                sum += (x.x + x.y) + (x.z + x.w);
              }
              if (g.out) {
                const float mean = wave_sum16(sum) * (1.f / 192.f);
                float qs = 0.f;
#pragma unroll
                for (int e = 0; e < 12; ++e) { v[e] -= mean; qs += v[e] * v[e]; }
                const float r = rsqrtf(wave_sum16(qs) * (1.f / 192.f) + g.eps);
                bf16x4 hq[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                  f32x4 t = {v[4 * i] * r, v[4 * i + 1] * r, v[4 * i + 2] * r, v[4 * i + 3] * r};
                  hq[i] = pack4(t);
                }
                bf16x4* hp = (bf16x4*)(g.out + (size_t)m * g.ldo);
#pragma unroll
                for (int i = 0; i < 3; ++i) xp[16 * i + c] = xs[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) hp[16 * i + c] = hq[i];
                if (c == 0) g.rstd_out[m] = r;
              } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) xp[16 * i + c] = xs[i];
              }
            } else {
              // LayerNorm backward behind a dgrad: v = dxhat row (affine already folded into the weight),
              // dX += rstd * (v - mean(v) - xhat * mean(v * xhat)); dXb = bf16(dX)
              float h[12];
              float s1 = 0.f, s2 = 0.f;
              const bf16x4* hp = (const bf16x4*)(g.mul + (size_t)m * g.ldm);
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                const bf16x4 t = hp[16 * i + c];
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[4 * i + e] = (float)t[e]; s1 += v[4 * i + e]; s2 += v[4 * i + e] * h[4 * i + e]; }
              }
              const float c1 = wave_sum16(s1) * (1.f / 192.f), c2 = wave_sum16(s2) * (1.f / 192.f);
              const float r = g.pos[m];
              bf16x4* bp = (bf16x4*)(g.out + (size_t)m * g.ldo);
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                float4 x;
                if (g.xprev) {                      // bf16 residual gradient in, bf16 out: the fp32 dX is not touched
                  const bf16x4 pb = ((const bf16x4*)(g.xprev + (size_t)m * 192))[16 * i + c];
                  x = (g.tokens <= 0 || (m % g.tokens) == 0) ? make_float4((float)pb[0], (float)pb[1], (float)pb[2], (float)pb[3])
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                  x = xp[16 * i + c];
                }
                x.x += r * (v[4 * i] - c1 - h[4 * i] * c2);
                x.y += r * (v[4 * i + 1] - c1 - h[4 * i + 1] * c2);
                x.z += r * (v[4 * i + 2] - c1 - h[4 * i + 2] * c2);
                x.w += r * (v[4 * i + 3] - c1 - h[4 * i + 3] * c2);
                if (!g.xprev) xp[16 * i + c] = x;
                f32x4 t = {x.x, x.y, x.z, x.w};
                bp[16 * i + c] = pack4(t);
              }
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < (ROWWISE ? 0 : (BM * 24 + NT - 1) / NT); ++q) {
        const int c = tid + q * NT;
        const int row = c / 24, ch = c - row * 24;
        const int m = mbase + row;
        if (((BM * 24) % NT == 0 || c < BM * 24) && m < g.M) {
          const bf16x8 pv = *(const bf16x8*)(Cs + row * CSTR + ch * 8);
          const size_t o = (size_t)m * g.ldo + n0 + ch * 8;
          if (EPI == EPI_BF16) {
            *(bf16x8*)(g.out + o) = pv;
          } else if (EPI == EPI_GELU) {
            bf16x8 av, dv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float ga, gd;
              gelu_and_grad((float)pv[e], ga, gd);
              av[e] = (bf16)ga; dv[e] = (bf16)gd;
            }
            *(bf16x8*)(g.out + o) = av;
            if (g.out2) *(bf16x8*)(g.out2 + o) = dv;
          } else {   // EPI_MUL
            const bf16x8 mv = *(const bf16x8*)(g.mul + (size_t)m * g.ldm + n0 + ch * 8);
            bf16x8 rv2;
#pragma unroll
            for (int e = 0; e < 8; ++e) rv2[e] = (bf16)((float)pv[e] * (float)mv[e]);
            *(bf16x8*)(g.out + o) = rv2;
          }
        }
      }
      barrier_lds();
    }
  };

  if (TWO_SETS) {
    for (int t = 0; t < ntile; t += 2) {
      if (t + 1 < ntile) lstore(1, rvA);           // tile t+1 -> buffer 1 (its readers finished at the last barrier)
      if (t + 3 < ntile) gload(tile0 + t + 3, rvA);
      tile_body(t, 0);
      if (t + 1 < ntile) {
        if (t + 2 < ntile) lstore(0, rvB);         // tile t+2 -> buffer 0
        if (t + 4 < ntile) gload(tile0 + t + 4, rvB);
        tile_body(t + 1, 1);
      }
    }
  } else {
    for (int t = 0; t < ntile; ++t) {
      if (t + 1 < ntile) lstore((t & 1) ^ 1, rvA);
      if (t + 2 < ntile) gload(tile0 + t + 2, rvA);
      tile_body(t, t & 1);
    }
  }
}

// CUs a weight-stationary launch sizes its grid for (256 = the whole chip).  The two-stream forward halves it so that
// the kernels of the two half-batch chains co-reside instead of queueing behind each other.
static thread_local int g_cu_budget = 256;      // per host thread: one thread drives one device

template <int KS, int WK, int BM>
int launch_ws(const GemmArgs& g0, int epi, hipStream_t st) {
  GemmArgs g = g0;
  g.n_tiles = g.N / 192;                                   // column chunks
  const int tiles_m = (g.M + BM - 1) / BM;
  const int wg_per_cu = WK == 1 ? 2 : 1;
  int pmax = (g_cu_budget * wg_per_cu) / g.n_tiles;
  pmax = pmax < 8 ? 8 : pmax / 8 * 8;
  int tpw = (tiles_m + pmax - 1) / pmax;                   // tiles per workgroup
  int P = (tiles_m + tpw - 1) / tpw;
  P = (P + 7) / 8 * 8;                                     // whole groups of 8 ids (one per XCD)
  constexpr int K = KS * 32 * WK;
  const size_t lds = (size_t)2 * BM * (K + 16) * sizeof(bf16) + (WK == 2 ? (size_t)2 * 8 * (BM / 16 / WK) * 3 * 64 * sizeof(f32x4) : 0) +
                     (size_t)BM * (192 + 8) * sizeof(bf16)
      ;
  dim3 grid(P * g.n_tiles), block(256 * WK);
#define LAUNCHW(E)                                                                                        \
  case E: {                                                                                               \
    if (!set_max_lds((const void*)gemm_ws_kernel<KS, WK, BM, E>, lds)) {                                  \
      rovit_set_error("gemm_ws: cannot raise the LDS limit");                                             \
      return ROVIT_ERR_LAUNCH;                                                                            \
    }                                                                                                     \
    hipLaunchKernelGGL((gemm_ws_kernel<KS, WK, BM, E>), grid, block, lds, st, g, tpw, tiles_m);           \
    break;                                                                                                \
  }
  switch (epi) {
    LAUNCHW(EPI_BF16) LAUNCHW(EPI_GELU) LAUNCHW(EPI_RESID) LAUNCHW(EPI_MUL) LAUNCHW(EPI_PATCH) LAUNCHW(EPI_RESID_LN) LAUNCHW(EPI_LNBWD)
    case EPI_PATCH_IMG:
      if constexpr (KS * WK * 32 == 768) {
        if (!set_max_lds((const void*)gemm_ws_kernel<KS, WK, BM, EPI_PATCH_IMG>, lds)) { rovit_set_error("gemm_ws: cannot raise the LDS limit"); return ROVIT_ERR_LAUNCH; }
        hipLaunchKernelGGL((gemm_ws_kernel<KS, WK, BM, EPI_PATCH_IMG>), grid, block, lds, st, g, tpw, tiles_m);
        break;
      }
      rovit_set_error("gemm_ws: the image-gather epilogue needs K = 768"); return ROVIT_ERR_SHAPE;
    default: rovit_set_error("gemm_ws: unknown epilogue %d", epi); return ROVIT_ERR_SHAPE;
  }
#undef LAUNCHW
  ROVIT_CHECK_LAUNCH("gemm_ws_kernel");
  return ROVIT_OK;
}

// ------------------------------------------------------------------------------------------------------
// K = 192 weight-stationary GEMM with the A tiles brought in by LDS-DMA (global_load_lds_dwordx4) into a 3-slot
// ring: no staging registers, so TWO 24 KB tiles per workgroup are in flight while a third is consumed.
// (The register-staged kernel above can keep only one tile per workgroup in flight, which caps a CU at ~25 GB/s
// of loads: its load/LDS/barrier skeleton alone costs ~16 us on the qkv shape.)
//   * LDS-DMA writes lane-linear (wave-uniform base + 16 B x lane), so rows are unpadded 384-byte rows and the
//     bank-conflict fix is an XOR swizzle of the 16-byte chunk index: physical chunk x of row r holds logical
//     chunk (x & ~7) | ((x & 7) ^ ((r >> 1) & 7)); applied on the SOURCE address of the DMA and on the fragment
//     read (conflict-free for ds_read_b128 fragment reads, checked by brute force over the b128 lane groups).
//   * completion: a wave's vmcnt covers its own DMA pieces and its epilogue stores, in issue order.  Per tile the
//     order is [DMA(t+2)] [stores(t)], so "tile t+1 landed" == at most SMAX + 6 younger ops outstanding.
//   * the staged-output tile aliases the ring slot that was just consumed (same XOR idea against write conflicts).
// Only for epilogues that issue no global LOADS inside the loop: a compiler-visible load would be waited for with
// vmcnt(0) and drain the DMA queue.  BF16 and GELU need none; EPI_MUL (fc2 dgrad x gelu') brings its (64 x 192) tile
// of the elementwise factor through a SECOND 3-slot ring by DMA as well (147 KB of LDS, one workgroup per CU, the
// same 2 x 49 KB in flight per CU): 51.8 -> 39.3 us on the fc2-dgrad shape, 4.4 TB/s.
// ------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_ws_dma_kernel(const GemmArgs g, int tiles_per_wg, int n_tiles_m) {
  constexpr int KS = 6, K = 192, BM = 64, TM = 4, NT = 256;
  constexpr int SLOT = BM * K;                              // bf16 elements per ring slot
  constexpr int SMAX = (EPI == EPI_GELU) ? 12 : 6;          // global stores one thread issues per tile
  constexpr bool HAS_MASK = (EPI == EPI_MUL);               // second ring: the (64 x 192) tile of the elementwise factor
  constexpr int PIECES = HAS_MASK ? 12 : 6;                 // DMA instructions one wave issues per tile
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];   // ring[3][64][192]; ONE array (see cdna guide)

  const int nchunks = g.n_tiles;
  const int bid = blockIdx.x;
  const int chunk = (bid >> 3) % nchunks;
  const int p = (bid / (8 * nchunks)) * 8 + (bid & 7);
  const int tile0 = p * tiles_per_wg;
  int ntile = n_tiles_m - tile0;
  ntile = ntile > tiles_per_wg ? tiles_per_wg : ntile;
  if (ntile <= 0) return;
  const int n0 = chunk * 192;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave;
  const int l15 = lane & 15, lg = lane >> 4;

  // DMA pieces of this wave: piece j = wave + 4 i covers linear chunks 64 j + lane of the slot
  int d_row[6], d_col[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = 64 * (wave + 4 * i) + lane;
    const int r = c / 24, x = c - r * 24;
    d_row[i] = r;
    d_col[i] = ((x & ~7) | ((x & 7) ^ ((r >> 1) & 7))) * 8;
  }
  auto dma = [&](int tile, int slot) {
    const int row0 = tile * BM;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int gr = row0 + d_row[i];
      gr = gr < g.M ? gr : g.M - 1;
      const bf16* src = g.A + (size_t)gr * g.lda + d_col[i];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + slot * SLOT + (wave + 4 * i) * 512), 16, 0, 0);
    }
    if (HAS_MASK) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        int gr = row0 + d_row[i];
        gr = gr < g.M ? gr : g.M - 1;
        const bf16* src = g.mul + (size_t)gr * g.ldm + n0 + d_col[i];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + (3 + slot) * SLOT + (wave + 4 * i) * 512), 16, 0, 0);
      }
    }
  };
  dma(tile0, 0);
  if (ntile > 1 && !GEMM_DBG(g, 4)) dma(tile0 + 1, 1);
  if (ntile > 2 && !GEMM_DBG(g, 4)) dma(tile0 + 2, 2);

  // stationary W fragments + bias (loaded once; nothing else is loaded from global memory inside the loop)
  bf16x8 wf[3][KS];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[j][ks] = *(const bf16x8*)(g.W + (size_t)(n0 + 48 * wn + 16 * j + l15) * g.ldw + ks * 32 + lg * 8);
  f32x4 bias4[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
      const float4 bb = *(const float4*)(g.bias + n0 + 48 * wn + 16 * j + 4 * lg);
      bias4[j] = (f32x4){bb.x, bb.y, bb.z, bb.w};
    }
  }
  // swizzled fragment-read offsets: logical chunk 4 ks + lg of row (16 i + l15)
  const int swz = (l15 >> 1) & 7;
  const int xo0 = ((0 + lg) ^ swz) * 8, xo1 = ((4 + lg) ^ swz) * 8;
  const int frag_row = l15 * K;
  // staged-output tile (aliases the consumed slot): 8-byte write of this lane, 16-byte read-back chunks
  const int cw_row = l15, cw_swz = l15 & 7;

  for (int t = 0; t < ntile; ++t) {
    const int slot = t % 3;
    // tile t has landed once at most (DMA(t+1..) + this tile's predecessors' stores) are outstanding
    if (t + 2 < ntile) {
      if (t == 0) wait_vmcnt<2 * PIECES>(); else wait_vmcnt<PIECES + SMAX>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();                     // every wave's pieces of tile t are in; tile t-1 fully retired
    if (t >= 1 && t + 2 < ntile && !GEMM_DBG(g, 4)) dma(tile0 + t + 2, (t + 2) % 3);     // refill the slot tile t-1 used

    f32x4 acc[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = bias4[j];
    const bf16* Ac = lds + slot * SLOT + frag_row;
    if (!GEMM_DBG(g, 2))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(Ac + i * 16 * K + (ks >> 1) * 64 + ((ks & 1) ? xo1 : xo0));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(wf[j][ks], af[i], acc[i][j]);     // D[n][m]
    }
    barrier_lds();                                    // all fragment reads of this slot are done: reuse it for C
    bf16* Cs = lds + slot * SLOT;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ch = 6 * wn + 2 * j + (lg >> 1);
        const int phys = (ch & ~7) | ((ch & 7) ^ cw_swz);
        *(bf16x4*)(Cs + (i * 16 + cw_row) * K + phys * 8 + 4 * (lg & 1)) = pack4(acc[i][j]);
      }
    barrier_lds();
    const int mbase = GEMM_DBG(g, 1) ? g.M : (tile0 + t) * BM;
#pragma unroll
    for (int q = 0; q < BM * 24 / NT; ++q) {
      const int c = tid + q * NT;
      const int row = c / 24, ch = c - row * 24;
      const int m = mbase + row;
      if (m < g.M) {
        const int phys = (ch & ~7) | ((ch & 7) ^ (row & 7));
        const bf16x8 pv = *(const bf16x8*)(Cs + row * K + phys * 8);
        const size_t o = (size_t)m * g.ldo + n0 + ch * 8;
        if (EPI == EPI_BF16 || (EPI == EPI_GELU && GEMM_DBG(g, 8))) {
          *(bf16x8*)(g.out + o) = pv;
          if (EPI == EPI_GELU && g.out2) *(bf16x8*)(g.out2 + o) = pv;
        } else if (EPI == EPI_MUL) {
          const int mphys = (ch & ~7) | ((ch & 7) ^ ((row >> 1) & 7));       // same swizzle the DMA wrote with
          const bf16x8 mv = *(const bf16x8*)(lds + (3 + slot) * SLOT + row * K + mphys * 8);
          bf16x8 rv2;
#pragma unroll
          for (int e = 0; e < 8; ++e) rv2[e] = (bf16)((float)pv[e] * (float)mv[e]);
          *(bf16x8*)(g.out + o) = rv2;
        } else {
          bf16x8 av, dv;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float ga, gd;
            gelu_and_grad((float)pv[e], ga, gd);
            av[e] = (bf16)ga; dv[e] = (bf16)gd;
          }
          *(bf16x8*)(g.out + o) = av;
          if (g.out2) *(bf16x8*)(g.out2 + o) = dv;
        }
      }
    }
    // no barrier here: the next iteration's top barrier retires this tile before its slot is refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int EPI>
int launch_ws_dma(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  g.n_tiles = g.N / 192;
  constexpr int BM = 64;
  const int tiles_m = (g.M + BM - 1) / BM;
  int pmax = (EPI == EPI_MUL ? 1 : 2) * g_cu_budget / g.n_tiles;      // workgroups per CU that fit the LDS ring(s)
  pmax = pmax < 8 ? 8 : pmax / 8 * 8;
  const int tpw = (tiles_m + pmax - 1) / pmax;
  int P = (tiles_m + tpw - 1) / tpw;
  P = (P + 7) / 8 * 8;
  const size_t lds = (size_t)(EPI == EPI_MUL ? 6 : 3) * BM * 192 * sizeof(bf16);
  ROVIT_CHECK_ARG(set_max_lds((const void*)gemm_ws_dma_kernel<EPI>, lds), ROVIT_ERR_LAUNCH, "gemm_ws_dma: cannot raise the LDS limit");
  hipLaunchKernelGGL((gemm_ws_dma_kernel<EPI>), dim3(P * g.n_tiles), dim3(256), lds, st, g, tpw, tiles_m);
  ROVIT_CHECK_LAUNCH("gemm_ws_dma_kernel");
  return ROVIT_OK;
}

// ------------------------------------------------------------------------------------------------------
// MLP backward, first half, with gelu' RECOMPUTED instead of read (round 2):
//     dpre[M,768] = (dY[M,192] W2T[768,192]^T)  *  gelu'( bf16( xhat2[M,192] W1[768,192]^T + b1 ) )
// The forward no longer writes gelu'(pre) (77 MB per block at batch 256) and this kernel reads the 19 MB of xhat2
// instead of those 77 MB: 174 -> 115 MB per launch, and the forward's fc1 GEMM drops from 174 to 97 MB.
// Same skeleton as gemm_ws_dma_kernel: both operand tiles (64 x 192) come in by LDS-DMA through two 3-slot rings
// (144 KB, one workgroup per CU), the weights are register-stationary.  TWELVE waves each own 16 output columns of
// the workgroup's 192-column chunk for BOTH products (2 x 24 registers of weights), so a lane ends up holding pre and
// dY.W2T for the same 4 columns of a row and the product needs no exchange.  pre is rounded to bf16 before gelu'
// because that is the value the forward's GELU saw (its tile is staged in bf16).
// ------------------------------------------------------------------------------------------------------
struct MlpBwdArgs {
  const bf16* dY; int ldy;        // (M,192) gradient w.r.t. the MLP output
  const bf16* H; int ldh;         // (M,192) xhat2, the fc1 input
  const bf16* W2T;                // (768,192) fc2 weight transposed: row n = d(out[:])/d(act[n])
  const bf16* W1;                 // (768,192) fc1 weight with the norm2 affine folded in
  const float* b1;                // (768) fc1 bias (folded)
  bf16* out; int ldo;             // (M,768) dpre
  int M, n_tiles;
};

// NW waves each own 192 / NW output columns of the workgroup's chunk for BOTH products; BM rows per tile.
//   <64, 12>: one 12-wave workgroup per CU (144 KB of LDS);  <32, 6>: two 6-wave workgroups per CU (72 KB each), whose
//   phases (DMA wait / MFMA / gelu' / store) interleave -- the epilogue is VALU-heavy (one exp2 + one rcp per element).
template <int BM, int NW>
__global__ __launch_bounds__(NW * 64, 3) void mlp_bwd_dma_kernel(const MlpBwdArgs g, int tiles_per_wg, int n_tiles_m) {
  constexpr int KS = 6, K = 192, TM = BM / 16, NT = NW * 64;
  constexpr int JT = 192 / NW / 16;                         // 16-column tiles per wave
  constexpr int SLOT = BM * K;
  constexpr int NPIECE = BM * 24 / 64;                      // 1 KB DMA pieces per tile and ring
  constexpr int PW = NPIECE / NW;                           // ... per wave
  constexpr int PIECES = 2 * PW;                            // DMA instructions one wave issues per tile (both rings)
  constexpr int SMAX = BM * 24 / NT;                        // global stores one thread issues per tile
  static_assert(NPIECE % NW == 0 && (BM * 24) % NT == 0 && 192 % (NW * 16) == 0, "tile must split evenly");
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];   // ring[6][BM][192]: slots 0-2 dY, 3-5 xhat2; ONE array

  const int nchunks = g.n_tiles;
  const int bid = blockIdx.x;
  const int chunk = (bid >> 3) % nchunks;                   // the chunks of one row range share an XCD (ids b, b+8, ...)
  const int p = (bid / (8 * nchunks)) * 8 + (bid & 7);
  const int tile0 = p * tiles_per_wg;
  int ntile = n_tiles_m - tile0;
  ntile = ntile > tiles_per_wg ? tiles_per_wg : ntile;
  if (ntile <= 0) return;
  const int n0 = chunk * 192;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;

  // DMA pieces of this wave: piece j = wave + NW i covers linear 16-byte chunks 64 j + lane of a slot
  int d_row[PW], d_col[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int c = 64 * (wave + NW * i) + lane;
    const int r = c / 24, x = c - r * 24;
    d_row[i] = r;
    d_col[i] = ((x & ~7) | ((x & 7) ^ ((r >> 1) & 7))) * 8;
  }
  auto dma = [&](int tile, int slot) {
    const int row0 = tile * BM;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int gr = row0 + d_row[i];
      gr = gr < g.M ? gr : g.M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.dY + (size_t)gr * g.ldy + d_col[i]),
                                       (__attribute__((address_space(3))) void*)(lds + slot * SLOT + (wave + NW * i) * 512), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      int gr = row0 + d_row[i];
      gr = gr < g.M ? gr : g.M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.H + (size_t)gr * g.ldh + d_col[i]),
                                       (__attribute__((address_space(3))) void*)(lds + (3 + slot) * SLOT + (wave + NW * i) * 512), 16, 0, 0);
    }
  };
  dma(tile0, 0);
  if (ntile > 1) dma(tile0 + 1, 1);
  if (ntile > 2) dma(tile0 + 2, 2);

  // stationary fragments of both weights: output column n0 + 16 (JT wave + j) + l15, k = 32 ks + 8 lg
  bf16x8 w2[JT][KS], w1[JT][KS];
  f32x4 bias4[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const int col = n0 + 16 * (JT * wave + j);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w2[j][ks] = *(const bf16x8*)(g.W2T + (size_t)(col + l15) * K + ks * 32 + lg * 8);
      w1[j][ks] = *(const bf16x8*)(g.W1 + (size_t)(col + l15) * K + ks * 32 + lg * 8);
    }
    const float4 bb = *(const float4*)(g.b1 + col + 4 * lg);
    bias4[j] = (f32x4){bb.x, bb.y, bb.z, bb.w};
  }
  const int swz = (l15 >> 1) & 7;
  const int xo0 = ((0 + lg) ^ swz) * 8, xo1 = ((4 + lg) ^ swz) * 8;
  const int frag_row = l15 * K;

  for (int t = 0; t < ntile; ++t) {
    const int slot = t % 3;
    // tile t has landed once at most (DMA(t+1) + the stores of tile t-1) are outstanding (issue order per iteration:
    // [DMA(t+2)] [stores(t)]); only the LAST tile of a launch can be partial, and it waits for everything
    if (t + 2 < ntile) {
      if (t == 0) wait_vmcnt<2 * PIECES>(); else wait_vmcnt<PIECES + SMAX>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (t >= 1 && t + 2 < ntile) dma(tile0 + t + 2, (t + 2) % 3);

    f32x4 pre[TM][JT], dac[TM][JT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j) { pre[i][j] = bias4[j]; dac[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const bf16* Ya = lds + slot * SLOT + frag_row;
    const bf16* Ha = lds + (3 + slot) * SLOT + frag_row;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int off = (ks >> 1) * 64 + ((ks & 1) ? xo1 : xo0);
      bf16x8 fy[TM], fh[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) { fh[i] = *(const bf16x8*)(Ha + i * 16 * K + off); fy[i] = *(const bf16x8*)(Ya + i * 16 * K + off); }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) { pre[i][j] = mfma16(w1[j][ks], fh[i], pre[i][j]); dac[i][j] = mfma16(w2[j][ks], fy[i], dac[i][j]); }   // D[n][m]
    }
    barrier_lds();                                    // all fragment reads of both slots are done: reuse the dY slot for the output tile
    bf16* Cs = lds + slot * SLOT;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = dac[i][j][r] * gelu_grad((float)(bf16)pre[i][j][r]);   // GELU saw the bf16-staged pre-activation
        const int row = i * 16 + l15;
        const int ch = 2 * (JT * wave + j) + (lg >> 1);
        const int phys = (ch & ~7) | ((ch & 7) ^ (row & 7));
        *(bf16x4*)(Cs + row * K + phys * 8 + 4 * (lg & 1)) = pack4(o);
      }
    barrier_lds();
    const int mbase = (tile0 + t) * BM;
#pragma unroll
    for (int q = 0; q < SMAX; ++q) {
      const int c = tid + q * NT;
      const int row = c / 24, ch = c - row * 24;
      const int m = mbase + row;
      if (m < g.M) {
        const int phys = (ch & ~7) | ((ch & 7) ^ (row & 7));
        *(bf16x8*)(g.out + (size_t)m * g.ldo + n0 + ch * 8) = *(const bf16x8*)(Cs + row * K + phys * 8);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next top barrier retires this tile before its slot is refilled
  }
}

template <int BM, int NW>
int launch_mlp_bwd(const MlpBwdArgs& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM;
  int pmax = g_cu_budget * (768 / (NW * 64)) / g.n_tiles;     // workgroups per CU that fit the 6-slot ring
  pmax = pmax < 8 ? 8 : pmax / 8 * 8;
  const int tpw = (tiles_m + pmax - 1) / pmax;
  int P = (tiles_m + tpw - 1) / tpw;
  P = (P + 7) / 8 * 8;
  const size_t lds = (size_t)6 * BM * 192 * sizeof(bf16);
  ROVIT_CHECK_ARG(set_max_lds((const void*)mlp_bwd_dma_kernel<BM, NW>, lds), ROVIT_ERR_LAUNCH, "gemm_mlp_bwd: cannot raise the LDS limit");
  hipLaunchKernelGGL((mlp_bwd_dma_kernel<BM, NW>), dim3(P * g.n_tiles), dim3(NW * 64), lds, st, g, tpw, tiles_m);
  ROVIT_CHECK_LAUNCH("mlp_bwd_dma_kernel");
  return ROVIT_OK;
}

// ------------------------------------------------------------------------------------------------------
// K = 576 / 768, N = 192 GEMMs with a row-wise LayerNorm epilogue (fc2 forward + norm, fc1 / qkv dgrad + norm
// backward), A tiles by LDS-DMA.  The register-staged kernel above splits K over wave pairs and needs an LDS
// exchange buffer next to its double-buffered A tiles; here TWELVE waves each own 16 output columns over the whole
// K (W fragments stationary: 96 / 72 registers), so there is no exchange and the 160 KB of LDS hold a 3-slot ring
// of (32 x K) tiles.  The epilogue's own global loads (residual rows, xhat, rstd) are issued at the TOP of the
// iteration so that their latency hides behind the MFMA phase (see the note at the loads about what the compiler's
// wait for them costs).  Used for K = 576 (qkv dgrad + norm1 backward: step 6.37 -> 6.27 ms); the K = 768
// instantiation spills at the 168-register budget of a 12-wave workgroup and stays off (ROVIT_KDMA768=1 enables it).
// ------------------------------------------------------------------------------------------------------
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int KT, int EPI>
__global__ __launch_bounds__(768, 1) void gemm_kdma_kernel(const GemmArgs g, int tiles_per_wg, int n_tiles_m) {
  constexpr int K = 32 * KT, BM = 32, NT = 768;
  constexpr int CPRW = K / 8;                               // 16-byte chunks per row: 96 / 72
  constexpr int SLOT = BM * K;
  constexpr int PIECES = BM * CPRW / NT;                    // DMA instructions per wave and tile: 4 / 3
  constexpr int NSTORE = (EPI == EPI_RESID_LN) ? 7 : 6;     // global stores a row-pass wave issues per tile
  constexpr int CSTR = 192 + 8;
  static_assert(BM * CPRW % NT == 0, "tile must split into whole DMA pieces");
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];   // ring[3][32][K]

  const int p = blockIdx.x;
  const int tile0 = p * tiles_per_wg;
  int ntile = n_tiles_m - tile0;
  ntile = ntile > tiles_per_wg ? tiles_per_wg : ntile;
  if (ntile <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const bool row_wave = wave < 8;                           // waves 0..7 run the row pass (512 threads = 32 rows x 16 lanes)

  // per piece: (row << 16) | swizzled source column, one register each; the source address is rebuilt per tile from
  // 32-bit offsets (keeping 64-bit per-lane pointers alive across the loop is what spills at K = 768)
  unsigned pk[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int c = 64 * (wave + 12 * i) + lane;              // linear 16-byte chunk of the slot
    const int r = c / CPRW, x = c - r * CPRW;
    // physical chunk x of row r holds logical chunk x ^ (r & 7) (inside its group of 8): conflict-free fragment reads
    pk[i] = ((unsigned)r << 16) | (unsigned)(((x & ~7) | ((x & 7) ^ (r & 7))) * 8);
  }
  auto dma = [&](int tile, int slot) {
    const int row0 = tile * BM;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      unsigned q = pk[i];
      asm volatile("" : "+v"(q));                           // keep the address arithmetic inside the loop
      int gr = row0 + (int)(q >> 16);
      gr = gr < g.M ? gr : g.M - 1;
      const bf16* src = g.A + ((unsigned)gr * (unsigned)g.lda + (q & 0xffffu));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + slot * SLOT + (wave + 12 * i) * 512), 16, 0, 0);
    }
  };
  dma(tile0, 0);
  if (ntile > 1) dma(tile0 + 1, 1);
  if (ntile > 2) dma(tile0 + 2, 2);

  bf16x8 wf[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks) wf[ks] = *(const bf16x8*)(g.W + (size_t)(16 * wave + l15) * g.ldw + ks * 32 + lg * 8);
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
    const float4 bb = *(const float4*)(g.bias + 16 * wave + 4 * lg);
    bias4 = (f32x4){bb.x, bb.y, bb.z, bb.w};
  }
  const int prow = tid >> 4, pc = tid & 15;                 // row pass: row of the tile, lane inside the row
  wait_vmcnt<0>();                                          // prologue: W fragments (and the first tiles) are in

  for (int t = 0; t < ntile; ++t) {
    const int slot = t % 3;
    if (t >= 1) {
      if (t + 2 < ntile) { if (row_wave) wait_vmcnt<PIECES + NSTORE>(); else wait_vmcnt<PIECES>(); }
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    // epilogue inputs of THIS tile first (their latency hides behind the MFMA phase).  NB the compiler waits for them
    // with vmcnt(0) -- it does not order LDS-DMA against register loads -- so the DMA issued below is drained at the
    // epilogue: effectively ONE tile ahead in flight, like the register-staged kernel, but without the exchange
    // buffer and with the epilogue loads off the critical path.  (Hand-issued asm loads are not an option: the compiler
    // may copy their destination registers before the data has landed.)
    const int m = (tile0 + t) * BM + prow;
    const bool live = row_wave && m < g.M;
    const int mc = m < g.M ? m : g.M - 1;
    f32x4 xo0, xo1, xo2;
    u32x2_t hx0, hx1, hx2;
    float rr = 0.f;
    // 32-bit element offsets from the (uniform) base pointers: saddr + voffset addressing, no 64-bit pointers held in
    // VGPRs across the loop (at K = 768 those were what spilled)
    const unsigned xoff = (unsigned)mc * (unsigned)g.ldx + 4u * pc;
    float* xrow = g.xres + xoff;
    constexpr bool PREFETCH = KT <= 18;                     // K = 768: 96 registers of W leave no room to hold them over the MFMAs
    u32x2_t pv0, pv1, pv2;                                  // EPI_LNBWD with a bf16 residual gradient (g.xprev): its row instead of xo*
    auto load_inputs = [&]() {
      if (EPI == EPI_LNBWD && g.xprev) {
        const bf16* prow_p = g.xprev + ((unsigned)mc * 192u + 4u * pc);
        pv0 = *(const u32x2_t*)prow_p; pv1 = *(const u32x2_t*)(prow_p + 64); pv2 = *(const u32x2_t*)(prow_p + 128);
      } else {
        xo0 = *(const f32x4*)xrow; xo1 = *(const f32x4*)(xrow + 64); xo2 = *(const f32x4*)(xrow + 128);
      }
      if (EPI == EPI_LNBWD) {
        const bf16* hrow = g.mul + ((unsigned)mc * (unsigned)g.ldm + 4u * pc);
        hx0 = *(const u32x2_t*)hrow; hx1 = *(const u32x2_t*)(hrow + 64); hx2 = *(const u32x2_t*)(hrow + 128);
        rr = g.pos[mc];
      }
    };
    if (PREFETCH && row_wave) load_inputs();
    const bool dma_now = t >= 1 && t + 2 < ntile;
    if (dma_now) dma(tile0 + t + 2, (t + 2) % 3);

    f32x4 acc[2] = {bias4, bias4};
    const bf16* Ac = lds + slot * SLOT;
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      const int lc = ks * 4 + lg;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 16 + l15;
        const bf16x8 af = *(const bf16x8*)(Ac + row * K + ((lc & ~7) | ((lc & 7) ^ (row & 7))) * 8);
        acc[i] = mfma16(wf[ks], af, acc[i]);               // D[n][m]
      }
    }
    barrier_lds();                                          // every wave is done with the slot: reuse it for C
    bf16* Cs = lds + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) *(bf16x4*)(Cs + (i * 16 + l15) * CSTR + 16 * wave + 4 * lg) = pack4(acc[i]);
    barrier_lds();
    if (row_wave) {
      if (!PREFETCH) load_inputs();
      float v[12];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bf16x4 tv = *(const bf16x4*)(Cs + prow * CSTR + 64 * i + 4 * pc);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)tv[e];
      }
      f32x4 xs[3] = {xo0, xo1, xo2};
      if (EPI == EPI_LNBWD && g.xprev) {
        const u32x2_t pvs[3] = {pv0, pv1, pv2};
        const bool has_in = g.tokens <= 0 || (mc % g.tokens) == 0;       // tokens > 0: only every tokens-th row carries an incoming gradient
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const bf16x4 pb = __builtin_bit_cast(bf16x4, pvs[i]);
          xs[i] = has_in ? (f32x4){(float)pb[0], (float)pb[1], (float)pb[2], (float)pb[3]} : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      f32x4* xp = (f32x4*)(g.xres + (unsigned)mc * (unsigned)g.ldx);
      if (EPI == EPI_RESID_LN) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xs[i][e] += v[4 * i + e];
          sum += (xs[i][0] + xs[i][1]) + (xs[i][2] + xs[i][3]);
        }
        const float mean = wave_sum16(sum) * (1.f / 192.f);
        float qs = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = xs[i][e] - mean; qs += d * d; }
        const float r = rsqrtf(wave_sum16(qs) * (1.f / 192.f) + g.eps);
        if (live) {
          bf16x4* hp = (bf16x4*)(g.out + (unsigned)m * (unsigned)g.ldo);
#pragma unroll
          for (int i = 0; i < 3; ++i) xp[16 * i + pc] = xs[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            f32x4 tq = {(xs[i][0] - mean) * r, (xs[i][1] - mean) * r, (xs[i][2] - mean) * r, (xs[i][3] - mean) * r};
            hp[16 * i + pc] = pack4(tq);
          }
          if (pc == 0) g.rstd_out[m] = r;
        }
      } else {
        const u32x2_t hxs[3] = {hx0, hx1, hx2};
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const bf16x4 hb = __builtin_bit_cast(bf16x4, hxs[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1 += v[4 * i + e]; s2 += v[4 * i + e] * (float)hb[e]; }
        }
        const float c1 = wave_sum16(s1) * (1.f / 192.f), c2 = wave_sum16(s2) * (1.f / 192.f);
        if (live) {
          bf16x4* bp = (bf16x4*)(g.out + (unsigned)m * (unsigned)g.ldo);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const bf16x4 hb = __builtin_bit_cast(bf16x4, hxs[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[i][e] += rr * (v[4 * i + e] - c1 - (float)hb[e] * c2);
          }
          if (!g.xprev && !GEMM_DBG(g, 64)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xp[16 * i + pc] = xs[i];
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) bp[16 * i + pc] = pack4(xs[i]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the next top barrier retires this tile's LDS reads
  }
}

template <int KT, int EPI>
int launch_kdma(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  constexpr int BM = 32;
  const int tiles_m = (g.M + BM - 1) / BM;
  int pmax = g_cu_budget;                                   // one 12-wave workgroup per CU
  const int tpw = (tiles_m + pmax - 1) / pmax;
  const int P = (tiles_m + tpw - 1) / tpw;
  const size_t lds = (size_t)3 * BM * 32 * KT * sizeof(bf16);
  ROVIT_CHECK_ARG(set_max_lds((const void*)gemm_kdma_kernel<KT, EPI>, lds), ROVIT_ERR_LAUNCH, "gemm_kdma: cannot raise the LDS limit");
  hipLaunchKernelGGL((gemm_kdma_kernel<KT, EPI>), dim3(P), dim3(768), lds, st, g, tpw, tiles_m);
  ROVIT_CHECK_LAUNCH("gemm_kdma_kernel");
  return ROVIT_OK;
}
// the K = 768 instantiation needs 96 registers of stationary W per lane and spills at the 168-register budget of a
// 12-wave workgroup: off unless ROVIT_KDMA768=1
// (the K = 768 instantiation of gemm_kdma_kernel reads 590 KB of LDS fragments per tile and spills at the 168-register budget: slower, not built)
static bool kdma_enabled() { return ROVIT_KNOB(ROVIT_KNOB_KDMA, 1) != 0; }

template <int BM, int BN, int WM, int WN>
int launch_nt(const GemmArgs& g0, int epi, hipStream_t st) {
  GemmArgs g = g0;
  g.n_tiles = g.N / BN;
  const int m_tiles = (g.M + BM - 1) / BM;
  const int nwg = m_tiles * g.n_tiles;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(bf16);
  dim3 grid(nwg), block(WM * WN * 64);
#define LAUNCH(E)                                                                                         \
  case E: {                                                                                               \
    if (!set_max_lds((const void*)gemm_nt_kernel<BM, BN, WM, WN, E>, lds)) {                              \
      rovit_set_error("gemm_nt: cannot raise the LDS limit");                                             \
      return ROVIT_ERR_LAUNCH;                                                                            \
    }                                                                                                     \
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, E>), grid, block, lds, st, g);                     \
    break;                                                                                                \
  }
  switch (epi) {
    LAUNCH(EPI_BF16) LAUNCH(EPI_GELU) LAUNCH(EPI_RESID) LAUNCH(EPI_MUL) LAUNCH(EPI_PATCH)
    default: rovit_set_error("gemm_nt: unknown epilogue %d", epi); return ROVIT_ERR_SHAPE;
  }
#undef LAUNCH
  ROVIT_CHECK_LAUNCH("gemm_nt_kernel");
  return ROVIT_OK;
}

// ------------------------------------------------------------------------------------------------------
// wgrad: G[n][k] = sum_m dY[m][n] A[m][k] over one M-split, written to slab[split][n][k] (fp32);
// colsum[split][n] = sum_m dY[m][n] comes out of one extra MFMA against a tile of ones.
// Both operands are m-major, so their MFMA fragments are gathered with ds_read_b64_tr_b16 from row-major
// LDS tiles.  Contraction slot (group g, element j) of a 32-row step holds row 16*(j>>2) + 4*g + (j&3):
// the two 16-lane groups of a half-wave then read 8 consecutive LDS rows (conflict-free at a 224-byte stride).
// ------------------------------------------------------------------------------------------------------
constexpr int WG_MSTEP = 64;
// LDS row stride (bf16 elements) of a [rows][T] operand tile: T + 16 or T + 32 so that the stride is an ODD multiple of 32 bytes
// (conflict-free ds_read_b64_tr_b16: the two 16-lane groups of a half-wave read 8 consecutive rows)
constexpr int wg_stride(int t) { return ((t + 16) * 2 / 32) % 2 ? t + 16 : t + 32; }

// One weight-gradient problem G[N][K] = dY[M][N]^T A[M][K] of a launch; several problems that share M (the four linears of
// a transformer block) run in ONE launch: with 24 output tiles per M-split instead of 2..8, the chip is filled with
// 16-24 M-splits instead of 32-128, so the fp32 partial slabs (splits x N x K x 4 bytes, written here and read again by
// the reduce kernel) shrink from 75.6 to 28-42 MB per block, and three launch ramps disappear.
struct WgradProb {
  const bf16* dY; int ldy;
  const bf16* A; int lda;
  int N, K;
  float* slab;          // [splits][N][K]
  float* colsum;        // [splits][N]
  int k_tiles, tile0;   // tiles of this problem are [tile0, tile0 + n_tiles * k_tiles) of a split's tile list
  int a_blk, y_blk;     // operand stored CHUNK-MAJOR [cols / 32][M][32] (the one-launch MLP half's act / dpre, mlp_fused.hip) instead of row-major
};
constexpr int WG_MAXPROB = 4;
struct WgradArgs {
  WgradProb p[WG_MAXPROB];
  int nprob;
  int M;
  int splits, rows_per_split, tiles_per_split;
  int patch_tokens;     // >0 (single problem only): dY row of m is m + m/(T-1) + 1 (skip each image's cls row)
  const float* img;     // PATCH only: when set, the A operand (patch pixels, K = 768) is gathered from these fp32 NCHW images
#ifdef ROVIT_DEV
  int dbg;              // developer library only (ROVIT_KNOB_GEMM_DBG >> 4): bit 0 skip steady-state global loads, bit 1 skip MFMAs
#endif
};

// Output tile TN (columns of dY = rows n of G) x TK (columns of A = columns k of G) per workgroup; four waves as 2 (k) x 2 (n),
// wave tile TK/2 x TN/2.  Smaller tiles mean fewer M-splits for the same number of workgroups, i.e. fewer fp32 partial
// slabs to write and reduce (slab bytes = splits x N x K x 4), at the price of more operand re-reads through the XCD's
// L2 (every tile of a split streams the same rows; all tiles of a split run on one XCD).
// WN = waves along n (2: the four-wave workgroup above; 4: EIGHT waves as 2 (k) x 4 (n) for 192 x 192 tiles, round 3).
// Staging chunk c (16 bytes) of a step's [WG_MSTEP rows][W columns] operand tile when the operand is stored CHUNK-MAJOR
// ([cols / 32][M][32]): eight consecutive lanes take row r of two neighbouring 32-column chunks, so that they still WRITE 128
// contiguous bytes of the LDS row (conflict-free, as in the row-major mapping) while a wave READS, per chunk, the 64-byte pieces
// of 8 consecutive rows = 512 contiguous bytes.  (Odd chunk counts, W = 96: four lanes per row and chunk.)
template <int W>
__device__ __forceinline__ void blk_map(int c, int& row, int& col) {
  if constexpr ((W / 32) % 2 == 0) {
    row = (c >> 3) % WG_MSTEP;
    col = (2 * ((c >> 3) / WG_MSTEP) + ((c >> 2) & 1)) * 32 + (c & 3) * 8;
  } else {
    row = (c >> 2) % WG_MSTEP;
    col = ((c >> 2) / WG_MSTEP) * 32 + (c & 3) * 8;
  }
}

template <int TN, int TK, bool PATCH, bool IMG = false, int WN = 2>
__global__ __launch_bounds__(128 * WN) void wgrad_kernel(const WgradArgs g) {
  constexpr int NT = 128 * WN;
  constexpr int SY = wg_stride(TN), SA = wg_stride(TK);
  constexpr int CPRY = TN / 8, CPRA = TK / 8;           // 16-byte chunks per tile row
  constexpr int CHY = WG_MSTEP * CPRY / NT, CHA = WG_MSTEP * CPRA / NT;   // chunks per thread and step
  constexpr int JN = TN / (16 * WN), IK = TK / 32;      // 16 x 16 accumulator tiles per wave: IK (k) x JN (n)
  static_assert(TN % (16 * WN) == 0 && TK % 32 == 0 && (WG_MSTEP * CPRY) % NT == 0 && (WG_MSTEP * CPRA) % NT == 0, "tile must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) bf16 lds[2 * WG_MSTEP * (SY + SA)];
  bf16* Ys = lds;                                       // [2][MSTEP][SY]
  bf16* As = lds + 2 * WG_MSTEP * SY;                   // [2][MSTEP][SA]

  const int bid = xcd_remap(blockIdx.x, gridDim.x);            // all tiles of a split on one XCD
  const int split = bid / g.tiles_per_split;
  int tile = bid - split * g.tiles_per_split;
  if (split >= g.splits) return;
  int pi = 0;
#pragma unroll
  for (int j = 1; j < WG_MAXPROB; ++j)
    if (j < g.nprob && tile >= g.p[j].tile0) pi = j;
  // (a scalar copy of the selected problem: pi is workgroup-uniform)
  const WgradProb pr = g.p[pi];
  tile -= pr.tile0;
  const int ntile = tile / pr.k_tiles, ktile = tile - ntile * pr.k_tiles;
  const int n0 = ntile * TN, k0 = ktile * TK;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WN, wn = wave % WN;             // wave tile: k rows [TK/2*wk, +TK/2), n cols [TN/WN*wn, +TN/WN)
  const int l15 = lane & 15, lg = lane >> 4;

  // Per-thread staging chunks.  Keep the hot loop free of address arithmetic: one clamped row index and one
  // 64-bit multiply-add per chunk and step; the LDS offsets are loop constants.
  int y_row[CHY], y_lds[CHY], a_row[CHA], a_lds[CHA], a_col[CHA];
  const bf16* ybase[CHY];
  const bf16* abase[CHA];
  using RA = typename std::conditional<IMG, F8, bf16x8>::type;      // IMG: raw pixels wait in registers, packed at the LDS store
#pragma unroll
  for (int i = 0; i < CHY; ++i) {
    const int c = tid + i * NT;
    int col;
    if (pr.y_blk) {       // chunk-major source: see blk_map
      blk_map<TN>(c, y_row[i], col);
      ybase[i] = pr.dY + (size_t)((n0 + col) >> 5) * g.M * 32 + ((n0 + col) & 31);
    } else {
      y_row[i] = c / CPRY;
      col = (c - y_row[i] * CPRY) * 8;
      ybase[i] = pr.dY + n0 + col;
    }
    y_lds[i] = y_row[i] * SY + col;
  }
  const int ldy_eff = pr.y_blk ? 32 : pr.ldy, lda_eff = pr.a_blk ? 32 : pr.lda;
#pragma unroll
  for (int i = 0; i < CHA; ++i) {
    const int c = tid + i * NT;
    int col;
    if (IMG) {      // 32 neighbouring patches x both halves of a 16-pixel segment per wave (see gemm_ws_kernel)
      const int rest = c >> 6, row_hi = rest / (CPRA / 2), pair = rest - row_hi * (CPRA / 2);
      a_row[i] = (c & 31) + 32 * row_hi;
      col = (pair * 2 + ((c >> 5) & 1)) * 8;
    } else if (pr.a_blk) {
      blk_map<TK>(c, a_row[i], col);
    } else {
      a_row[i] = c / CPRA;
      col = (c - a_row[i] * CPRA) * 8;
    }
    a_lds[i] = a_row[i] * SA + col;
    a_col[i] = k0 + col;
    abase[i] = pr.a_blk ? pr.A + (size_t)((k0 + col) >> 5) * g.M * 32 + ((k0 + col) & 31) : pr.A + k0 + col;
  }
  const int m_last = g.M - 1;
  auto load = [&](int mbase, bf16x8* ry, RA* ra) {
#pragma unroll
    for (int i = 0; i < CHY; ++i) {
      int m = mbase + y_row[i];
      m = m < m_last ? m : m_last;                      // clamp (never branch around a load); tail rows are zeroed in store()
      const int yr = PATCH ? m + m / (g.patch_tokens - 1) + 1 : m;
      ry[i] = *(const bf16x8*)(ybase[i] + (size_t)yr * ldy_eff);
    }
#pragma unroll
    for (int i = 0; i < CHA; ++i) {
      int m = mbase + a_row[i];
      m = m < m_last ? m : m_last;
      if constexpr (IMG) ra[i] = load_f8(patch_src(g.img, m, a_col[i]));
      else ra[i] = *(const bf16x8*)(abase[i] + (size_t)m * lda_eff);
    }
  };
  auto tobf = [&](const RA& v) -> bf16x8 {
    if constexpr (IMG) return pack8(v.lo, v.hi);
    else return v;
  };
  auto store = [&](int buf, int mbase, const bf16x8* ry, const RA* ra) {
    bf16* yd = Ys + buf * WG_MSTEP * SY;
    bf16* ad = As + buf * WG_MSTEP * SA;
    if (mbase + WG_MSTEP <= m_end) {                    // wave-uniform: full step
#pragma unroll
      for (int i = 0; i < CHY; ++i) *(bf16x8*)(yd + y_lds[i]) = ry[i];
#pragma unroll
      for (int i = 0; i < CHA; ++i) *(bf16x8*)(ad + a_lds[i]) = tobf(ra[i]);
    } else {                                            // last, partial step of a split: zero the rows past m_end
#pragma unroll
      for (int i = 0; i < CHY; ++i) *(bf16x8*)(yd + y_lds[i]) = keep_if(ry[i], mbase + y_row[i] < m_end);
#pragma unroll
      for (int i = 0; i < CHA; ++i) *(bf16x8*)(ad + a_lds[i]) = keep_if(tobf(ra[i]), mbase + a_row[i] < m_end);
    }
  };

  f32x4 acc[IK][JN], accb[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IK; ++i) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (bf16)1.0f;
  // bias gradient: one extra MFMA per 16 columns against a tile of ones.  The waves that hold the same dY
  // fragments (same n-tile: 2 wk x k_tiles workgroups) share the column tiles between them.
  const int cs_owner = ktile * 2 + wk, cs_n = 2 * pr.k_tiles;
  bool cs_do[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) cs_do[j] = (j % cs_n) == cs_owner;

  const int nsteps = (m_end - m_begin + WG_MSTEP - 1) / WG_MSTEP;
  // transposed-read lane address inside a [rows][stride] tile: row 4*lg + (l15>>2), col 4*(l15&3)
  const bf16* Ybase = Ys + (4 * lg + (l15 >> 2)) * SY + 4 * (l15 & 3) + wn * (TN / WN);
  const bf16* Abase = As + (4 * lg + (l15 >> 2)) * SA + 4 * (l15 & 3) + wk * (TK / 2);
  auto compute = [&](int cur) {
    const bf16* Yc = Ybase + cur * WG_MSTEP * SY;
    const bf16* Ac = Abase + cur * WG_MSTEP * SA;
#pragma unroll
    for (int ms = 0; ms < WG_MSTEP / 32; ++ms) {
      bf16x8 fa[IK], fy[JN];
#pragma unroll
      for (int i = 0; i < IK; ++i) {
        const bf16* p = Ac + ms * 32 * SA + i * 16;
        fa[i] = cat4(lds_read_tr(p), lds_read_tr(p + 16 * SA));
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const bf16* q = Yc + ms * 32 * SY + j * 16;
        fy[j] = cat4(lds_read_tr(q), lds_read_tr(q + 16 * SY));
      }
#pragma unroll
      for (int i = 0; i < IK; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = mfma16(fa[i], fy[j], acc[i][j]);      // D[k][n]
#pragma unroll
      for (int j = 0; j < JN; ++j)
        if (cs_do[j]) accb[j] = mfma16(ones, fy[j], accb[j]);
    }
  };
  // Two register sets keep the global loads of steps s+1 and s+2 in flight while step s runs its MFMAs; the
  // barrier waits for LDS only, so those loads are not drained at it.
  bf16x8 ryA[CHY], ryB[CHY];
  RA raA[CHA], raB[CHA];
  load(m_begin, ryA, raA);
  store(0, m_begin, ryA, raA);
  load(m_begin + WG_MSTEP, ryA, raA);
  load(m_begin + 2 * WG_MSTEP, ryB, raB);
  barrier_lds();
  for (int s = 0; s < nsteps; s += 2) {
    const int mb = m_begin + s * WG_MSTEP;
    if (s + 1 < nsteps) store(1, mb + WG_MSTEP, ryA, raA);
    if (!GEMM_DBG(g, 1)) load(mb + 3 * WG_MSTEP, ryA, raA);
    if (!GEMM_DBG(g, 2)) compute(0);
    barrier_lds();
    if (s + 1 < nsteps) {
      if (s + 2 < nsteps) store(0, mb + 2 * WG_MSTEP, ryB, raB);
      if (!GEMM_DBG(g, 1)) load(mb + 4 * WG_MSTEP, ryB, raB);
      if (!GEMM_DBG(g, 2)) compute(1);
      barrier_lds();
    }
  }
  float* slab = pr.slab + (size_t)split * pr.N * pr.K;
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int n = n0 + wn * (TN / WN) + j * 16 + l15;
#pragma unroll
    for (int i = 0; i < IK; ++i) {
      const int k = k0 + wk * (TK / 2) + i * 16 + lg * 4;
      *(float4*)(slab + (size_t)n * pr.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    if (cs_do[j] && lg == 0) pr.colsum[(size_t)split * pr.N + n] = accb[j][0];
  }
}

// Sum the slabs of a weight gradient; optionally un-fold the LayerNorm affine that was folded into the weight:
//   W_f = W * gamma (per k), b_f = b + W beta   =>   dW = gamma * G + beta (x) db,  dgamma = sum_n W G,
//   dbeta = sum_n W db  (finished by the affine kernel, which needs the complete db).
struct ReduceBatch { RovitReduceDesc d[ROVIT_REDUCE_BATCH]; int first_block[ROVIT_REDUCE_BATCH + 1]; int n; };

// Plain problems: thread = 4 consecutive elements of the flattened (N, K) gradient, summed over the slabs in split order.
// Problems with a folded LayerNorm affine (gamma != NULL; N % 16 == 0, K % 64 == 0): a workgroup owns 16 rows x 64 columns,
// so that the un-folding (dW = gamma*G + beta (x) db, and the per-column sums dgamma = sum_n W G, dbeta = sum_n W db) happens
// in the same pass: the raw G never goes back to memory (round 1 wrote it to a scratch buffer and ran a second kernel over it).
// Column sums leave the workgroup as one partial per 16-row group, combined in row order by the finalize kernel
// (deterministic: no float atomics).
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch rb) {
  __shared__ float s_pg[16][64], s_pb[16][64];
  int i = 0;
  while (i + 1 < rb.n && (int)blockIdx.x >= rb.first_block[i + 1]) ++i;
  const RovitReduceDesc& d = rb.d[i];
  const float* slab = d.ws;
  const float* colsum = d.ws + (size_t)d.splits * d.N * d.K;
  const int local = (int)blockIdx.x - rb.first_block[i];
  if (!d.gamma) {
    const int q = local * 256 + threadIdx.x;
    const int total4 = d.N * d.K / 4;
    if (q < total4) {
      const float4* sp = (const float4*)slab + q;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int j = 0; j < d.splits; ++j) {
        const float4 v = sp[(size_t)j * total4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      ((float4*)d.dW)[q] = s;
    }
    if (q < d.N) {
      float cb = 0.f;
#pragma unroll 8
      for (int j = 0; j < d.splits; ++j) cb += colsum[(size_t)j * d.N + q];
      d.db[q] = cb;
    }
    return;
  }
  const int kb = d.K / 64;
  const int rblk = local / kb, kblk = local - rblk * kb;
  const int r = threadIdx.x >> 4, kq = threadIdx.x & 15;
  const int n = rblk * 16 + r, k = kblk * 64 + kq * 4;
  const size_t e = (size_t)n * d.K + k;
  const size_t stride = (size_t)d.N * d.K;
  float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
  float cb = 0.f;
#pragma unroll 8
  for (int j = 0; j < d.splits; ++j) {
    const float4 v = *(const float4*)(slab + (size_t)j * stride + e);
    G.x += v.x; G.y += v.y; G.z += v.z; G.w += v.w;
    cb += colsum[(size_t)j * d.N + n];
  }
  const float4 gam = *(const float4*)(d.gamma + k), bet = *(const float4*)(d.beta + k), w = *(const float4*)(d.W + e);
  *(float4*)(d.dW + e) = make_float4(fmaf(gam.x, G.x, bet.x * cb), fmaf(gam.y, G.y, bet.y * cb), fmaf(gam.z, G.z, bet.z * cb),
                                     fmaf(gam.w, G.w, bet.w * cb));
  if (kblk == 0 && kq == 0) d.db[n] = cb;
  *(float4*)&s_pg[r][kq * 4] = make_float4(w.x * G.x, w.y * G.y, w.z * G.z, w.w * G.w);
  *(float4*)&s_pb[r][kq * 4] = make_float4(w.x * cb, w.y * cb, w.z * cb, w.w * cb);
  __syncthreads();
  if (threadIdx.x < 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) { sg += s_pg[rr][threadIdx.x]; sb += s_pb[rr][threadIdx.x]; }
    const int slices = d.N / 16;
    float* part = d.g_scratch;                 // [2][N/16][K] partial column sums (the scratch holds N*K floats)
    part[(size_t)rblk * d.K + kblk * 64 + threadIdx.x] = sg;
    part[(size_t)(slices + rblk) * d.K + kblk * 64 + threadIdx.x] = sb;
  }
}

__global__ __launch_bounds__(256) void wgrad_affine_finalize_kernel(const ReduceBatch rb) {
  int i = 0;
  while (i + 1 < rb.n && (int)blockIdx.x >= rb.first_block[i + 1]) ++i;
  const RovitReduceDesc& d = rb.d[i];
  const int k = ((int)blockIdx.x - rb.first_block[i]) * 256 + threadIdx.x;
  if (k >= d.K) return;
  const float* part = d.g_scratch;
  const int slices = d.N / 16;
  float sg = 0.f, sb = 0.f;
#pragma unroll 4
  for (int sl = 0; sl < slices; ++sl) { sg += part[(size_t)sl * d.K + k]; sb += part[(size_t)(slices + sl) * d.K + k]; }
  d.dgamma[k] = sg;
  d.dbeta[k] = sb;
}

}  // namespace

void rovit_set_cu_budget(int cus) { g_cu_budget = cus < 8 ? 8 : (cus > 256 ? 256 : cus); }

extern "C" int rovit_gemm_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, int epi,
                             void* out, int ldo, void* out2, float* xres, int ldx, const void* mul, int ldm,
                             const float* pos, int tokens, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(A && W, ROVIT_ERR_NULL, "gemm_nt: null operand");
  ROVIT_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 96 == 0, ROVIT_ERR_SHAPE,
                  "gemm_nt: unsupported shape M=%d N=%d K=%d (K %% 64, N %% 96)", M, N, K);
  ROVIT_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && rovit_aligned16(A) && rovit_aligned16(W), ROVIT_ERR_ALIGN,
                  "gemm_nt: operands must be 16-byte aligned with ld %% 8 == 0");
  GemmArgs g{};
  g.A = (const bf16*)A; g.lda = lda; g.W = (const bf16*)W; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.bias = bias;
  g.out = (bf16*)out; g.ldo = ldo; g.out2 = (bf16*)out2; g.xres = xres; g.ldx = ldx; g.mul = (const bf16*)mul; g.ldm = ldm;
  g.pos = pos; g.tokens = tokens; GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0));
  const int tiled = epi & (ROVIT_GEMM_TILED_192 | ROVIT_GEMM_TILED_96);       // per-call: force the LDS-tiled kernels (tests of that path)
  epi &= 0xff;
  switch (epi) {
    case EPI_BF16: case EPI_GELU: ROVIT_CHECK_ARG(out && ldo % 4 == 0, ROVIT_ERR_NULL, "gemm_nt: bf16 output missing"); break;
    case EPI_RESID: ROVIT_CHECK_ARG(xres && ldx % 4 == 0, ROVIT_ERR_NULL, "gemm_nt: residual stream missing"); break;
    case EPI_MUL: ROVIT_CHECK_ARG(out && mul, ROVIT_ERR_NULL, "gemm_nt: multiplier missing"); break;
    case EPI_PATCH: ROVIT_CHECK_ARG(xres && pos && tokens > 1 && M % (tokens - 1) == 0, ROVIT_ERR_SHAPE, "gemm_nt: bad patch epilogue"); break;
    default: ROVIT_CHECK_ARG(false, ROVIT_ERR_SHAPE, "gemm_nt: unknown epilogue %d", epi);
  }
  if (!tiled && N % 192 == 0) {
    if (K == 192 && epi == EPI_BF16) return launch_ws_dma<EPI_BF16>(g, (hipStream_t)stream);
    if (K == 192 && epi == EPI_GELU) return launch_ws_dma<EPI_GELU>(g, (hipStream_t)stream);
    if (K == 192 && epi == EPI_MUL) return launch_ws_dma<EPI_MUL>(g, (hipStream_t)stream);
    if (K == 192) return launch_ws<6, 1, 64>(g, epi, (hipStream_t)stream);
    if (K == 576) return launch_ws<9, 2, 32>(g, epi, (hipStream_t)stream);
    if (K == 768) return launch_ws<12, 2, 32>(g, epi, (hipStream_t)stream);
  }
  if (!(tiled & ROVIT_GEMM_TILED_96) && N % 192 == 0) return launch_nt<128, 192, 2, 2>(g, epi, (hipStream_t)stream);
  return launch_nt<128, 96, 2, 2>(g, epi, (hipStream_t)stream);
}

// PatchEmbed forward without the im2col buffer: X[b*tokens + 1 + p][:] = patches(images)[b, p, :] W^T + bias + pos[1 + p]
// (conv k16 s16 of timm's PatchEmbed as a GEMM whose A tiles are gathered from the fp32 NCHW images and rounded to bf16 on
// the way into LDS, exactly what rovit_im2col + rovit_gemm_nt(EPI_PATCH) compute).
extern "C" int rovit_patch_embed_fwd(const float* images, const void* W, const float* bias, const float* pos, float* X, int batch, int tokens,
                                     rovit_stream_t stream) {
  ROVIT_CHECK_ARG(images && W && pos && X, ROVIT_ERR_NULL, "patch_embed_fwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && tokens == 197, ROVIT_ERR_SHAPE, "patch_embed_fwd: 224x224 images in 16x16 patches (197 tokens) only");
  ROVIT_CHECK_ARG(rovit_aligned16(images) && rovit_aligned16(W) && rovit_aligned16(X) && rovit_aligned16(pos), ROVIT_ERR_ALIGN,
                  "patch_embed_fwd: buffers must be 16-byte aligned");
  GemmArgs g{};
  g.A = nullptr; g.lda = 768; g.W = (const bf16*)W; g.ldw = 768; g.M = batch * 196; g.N = 192; g.K = 768; g.bias = bias;
  g.xres = X; g.ldx = 192; g.pos = pos; g.tokens = tokens; g.img = images; GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0));
  return launch_ws<12, 2, 32>(g, EPI_PATCH_IMG, (hipStream_t)stream);
}

#ifdef ROVIT_DEV      // round 2's memory mode (gelu' recomputed by the backward: 64 us against 40): developer library only
// dpre = (dY W2T^T) * gelu'(bf16(H W1^T + b1)): first half of the MLP backward with gelu' recomputed from xhat2
// (reference arithmetic: autograd of timm Mlp, fc2 then GELU then fc1)
extern "C" int rovit_gemm_mlp_bwd(const void* dY, int ldy, const void* H, int ldh, const void* W2T, const void* W1, const float* b1,
                                  int M, void* dpre, int ldo, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && H && W2T && W1 && b1 && dpre, ROVIT_ERR_NULL, "gemm_mlp_bwd: null pointer");
  ROVIT_CHECK_ARG(M > 0, ROVIT_ERR_SHAPE, "gemm_mlp_bwd: M must be positive");
  ROVIT_CHECK_ARG(ldy % 8 == 0 && ldh % 8 == 0 && ldo % 8 == 0 && rovit_aligned16(dY) && rovit_aligned16(H) && rovit_aligned16(W2T) &&
                      rovit_aligned16(W1) && rovit_aligned16(b1) && rovit_aligned16(dpre),
                  ROVIT_ERR_ALIGN, "gemm_mlp_bwd: operands must be 16-byte aligned with ld %% 8 == 0");
  MlpBwdArgs g{(const bf16*)dY, ldy, (const bf16*)H, ldh, (const bf16*)W2T, (const bf16*)W1, b1, (bf16*)dpre, ldo, M, 4};
  // one 12-wave workgroup per CU: 64.2 us at M = 50432; the <32, 6> form (two 6-wave workgroups per CU) needs 96 registers of
  // weights per lane and spills at the 168-register budget of 3 waves per SIMD (102 us): not instantiated
  return launch_mlp_bwd<64, 12>(g, (hipStream_t)stream);
}
#endif

extern "C" size_t rovit_wgrad_workspace_bytes(int N, int K, int splits) {
  return ((size_t)splits * N * K + (size_t)splits * N) * sizeof(float);
}

// Tile of the single-problem launch: 96 x 96 (developer library: ROVIT_KNOB_WGRAD_TILE = (tn << 16) | tk for A/B timing).
static void wgrad_tile_for(int N, int K, int* tn, int* tk) {
  const int knob = ROVIT_KNOB(ROVIT_KNOB_WGRAD_TILE, 0);
  int n = knob >> 16, k = knob & 0xffff;
  if (n <= 0 || k <= 0 || N % n || K % k) { n = 96; k = 96; }
  *tn = n; *tk = k;
}

extern "C" int rovit_wgrad_splits(int M, int N, int K) {
  int tn, tk;
  wgrad_tile_for(N, K, &tn, &tk);
  const int tiles = (N / tn) * (K / tk);
  const int target = ROVIT_KNOB(ROVIT_KNOB_WGRAD_WGS, 512);
  int s = (target + tiles - 1) / tiles;            // ~2 workgroups per CU: measured best trade against slab traffic
  s = (s + 7) / 8 * 8;
  const int max_s = (M + WG_MSTEP - 1) / WG_MSTEP;
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}

template <int TN, int TK, int WN = 2>
static void launch_wgrad(const WgradArgs& g0, hipStream_t st) {
  WgradArgs g = g0;
  int t = 0;
  for (int j = 0; j < g.nprob; ++j) {
    g.p[j].k_tiles = g.p[j].K / TK;
    g.p[j].tile0 = t;
    t += (g.p[j].N / TN) * g.p[j].k_tiles;
  }
  g.tiles_per_split = t;
  const int nwg = g.splits * t;
  if constexpr (WN != 2) {
    hipLaunchKernelGGL((wgrad_kernel<TN, TK, false, false, WN>), dim3(nwg), dim3(128 * WN), 0, st, g);
  } else {
    if (g.patch_tokens > 0 && g.img) hipLaunchKernelGGL((wgrad_kernel<TN, TK, true, true>), dim3(nwg), dim3(256), 0, st, g);
    else if (g.patch_tokens > 0) hipLaunchKernelGGL((wgrad_kernel<TN, TK, true>), dim3(nwg), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((wgrad_kernel<TN, TK, false>), dim3(nwg), dim3(256), 0, st, g);
  }
}

// slab/colsum live in `ws` (rovit_wgrad_workspace_bytes); results are produced by rovit_wgrad_reduce.
extern "C" int rovit_wgrad(const void* dY, int ldy, const void* A, int lda, int M, int N, int K, int splits, int patch_tokens,
                           float* ws, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && A && ws, ROVIT_ERR_NULL, "wgrad: null pointer");
  ROVIT_CHECK_ARG(M > 0 && N % 96 == 0 && K % 96 == 0 && splits > 0, ROVIT_ERR_SHAPE, "wgrad: unsupported shape N=%d K=%d", N, K);
  ROVIT_CHECK_ARG(ldy % 8 == 0 && lda % 8 == 0 && rovit_aligned16(dY) && rovit_aligned16(A), ROVIT_ERR_ALIGN, "wgrad: alignment");
  WgradArgs g{};
  g.nprob = 1;
  g.p[0] = WgradProb{(const bf16*)dY, ldy, (const bf16*)A, lda, N, K, ws, ws + (size_t)splits * N * K, 0, 0};
  g.M = M;
  g.splits = splits;
  g.rows_per_split = ((M + splits - 1) / splits + WG_MSTEP - 1) / WG_MSTEP * WG_MSTEP;
  g.patch_tokens = patch_tokens;
  GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0) >> 4);
  // a split whose first row is past M still writes zeros, so the reduce can sum every slab
  int tn, tk;
  wgrad_tile_for(N, K, &tn, &tk);
  hipStream_t st = (hipStream_t)stream;
  switch ((tn << 16) | tk) {
    case (96 << 16) | 96: launch_wgrad<96, 96>(g, st); break;
#ifdef ROVIT_DEV      // tile sweep of round 2 (profiles/r02_wgrad_tile_sweep.txt)
    case (64 << 16) | 96: launch_wgrad<64, 96>(g, st); break;
    case (96 << 16) | 64: launch_wgrad<96, 64>(g, st); break;
    case (64 << 16) | 64: launch_wgrad<64, 64>(g, st); break;
    case (96 << 16) | 192: launch_wgrad<96, 192>(g, st); break;
    case (192 << 16) | 96: launch_wgrad<192, 96>(g, st); break;
#endif
    default: rovit_set_error("wgrad: no kernel for tile %d x %d", tn, tk); return ROVIT_ERR_SHAPE;
  }
  ROVIT_CHECK_LAUNCH("wgrad_kernel");
  return ROVIT_OK;
}

// Weight gradient of the PatchEmbed projection without an im2col buffer: G[n][k] = sum over (image, patch) rows m of
// dY[token row of m][n] * pixel(m, k), pixels gathered from the fp32 NCHW images and rounded to bf16 on the way into LDS
// (what rovit_wgrad(..., patch_tokens) computes from the rovit_im2col buffer).  dY: bf16 (batch*tokens, N) token rows.
extern "C" int rovit_patch_embed_wgrad(const void* dY, int ldy, const float* images, int batch, int tokens, int N, int splits, float* ws,
                                       rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && images && ws, ROVIT_ERR_NULL, "patch_embed_wgrad: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && tokens == 197 && N % 96 == 0 && splits > 0, ROVIT_ERR_SHAPE, "patch_embed_wgrad: unsupported shape");
  ROVIT_CHECK_ARG(ldy % 8 == 0 && rovit_aligned16(dY) && rovit_aligned16(images), ROVIT_ERR_ALIGN, "patch_embed_wgrad: alignment");
  const int M = batch * (tokens - 1), K = 768;
  WgradArgs g{};
  g.nprob = 1;
  g.p[0] = WgradProb{(const bf16*)dY, ldy, nullptr, K, N, K, ws, ws + (size_t)splits * N * K, 0, 0};
  g.M = M;
  g.splits = splits;
  g.rows_per_split = ((M + splits - 1) / splits + WG_MSTEP - 1) / WG_MSTEP * WG_MSTEP;
  g.patch_tokens = tokens;
  g.img = images;
  GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0) >> 4);
  // 192-wide output tiles when N allows: the fp32 pixels (twice the bytes of a bf16 im2col row) are then read by one
  // workgroup per M-split instead of two (61.6 -> 56.3 us at batch 256)
  if (N % 192 == 0) launch_wgrad<192, 96>(g, (hipStream_t)stream);
  else launch_wgrad<96, 96>(g, (hipStream_t)stream);
  ROVIT_CHECK_LAUNCH("wgrad_kernel (patch, image gather)");
  return ROVIT_OK;
}

// Several weight gradients that share M in one launch (96 x 192 tiles, `splits` M-splits for all of them); the slabs of
// problem j live in descs[j].ws like rovit_wgrad's (rovit_wgrad_workspace_bytes(N, K, splits)).
int rovit_wgrad_batch(const RovitWgradDesc* descs, int n, int M, int splits, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(descs && n >= 1 && n <= WG_MAXPROB && M > 0 && splits > 0, ROVIT_ERR_SHAPE, "wgrad_batch: bad arguments");
  WgradArgs g{};
  g.nprob = n;
  for (int j = 0; j < n; ++j) {
    const RovitWgradDesc& d = descs[j];
    ROVIT_CHECK_ARG(d.dY && d.A && d.ws, ROVIT_ERR_NULL, "wgrad_batch: null pointer");
    ROVIT_CHECK_ARG(d.N % 96 == 0 && d.K % 192 == 0, ROVIT_ERR_SHAPE, "wgrad_batch: unsupported shape N=%d K=%d", d.N, d.K);
    ROVIT_CHECK_ARG(d.ldy % 8 == 0 && d.lda % 8 == 0 && rovit_aligned16(d.dY) && rovit_aligned16(d.A), ROVIT_ERR_ALIGN, "wgrad_batch: alignment");
    ROVIT_CHECK_ARG(!(d.a_blk && d.K % 32) && !(d.y_blk && d.N % 32), ROVIT_ERR_SHAPE, "wgrad_batch: chunk-major operands need 32-column chunks");
    g.p[j] = WgradProb{(const bf16*)d.dY, d.ldy, (const bf16*)d.A, d.lda, d.N, d.K, d.ws, d.ws + (size_t)splits * d.N * d.K, 0, 0, d.a_blk, d.y_blk};
  }
  g.M = M;
  g.splits = splits;
  g.rows_per_split = ((M + splits - 1) / splits + WG_MSTEP - 1) / WG_MSTEP * WG_MSTEP;
  GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0) >> 4);
  // a lone problem (the qkv weight gradient flushed at the end of a data-parallel block range): 96 x 96 tiles give twice the
  // workgroups for the same 16 M-splits (96 -> 192); every element still sums the same rows in the same order, so the
  // result is bit-identical to the merged launch's
  // 192 x 192 tiles with eight waves (round 3, the default): every row of dY and A that a workgroup stages feeds 192 output
  // columns instead of 64-96, so the L2 -> CU re-read traffic of the launch halves (the per-CU load path, not HBM, bounded the
  // smaller tiles: DESIGN.md "weight gradients").  12 tiles per M-split; 16 splits = 192 workgroups in the step (the other
  // stream's dgrad kernels keep the remaining CUs busy), 21 = 252 is the fastest standalone (73 us against 107 us for 64 x 192
  // tiles x 14 splits on the same box).
  if (n == 1 && g.p[0].K % 96 == 0) launch_wgrad<96, 96>(g, (hipStream_t)stream);
  else {                             // 192 x 192 tiles, eight waves: 12 tiles per split
    for (int j = 0; j < n; ++j) ROVIT_CHECK_ARG(descs[j].N % 192 == 0 && descs[j].K % 192 == 0, ROVIT_ERR_SHAPE, "wgrad_batch: N, K %% 192 for 192-wide tiles");
    launch_wgrad<192, 192, 4>(g, (hipStream_t)stream);
  }
  ROVIT_CHECK_LAUNCH("wgrad_kernel (batch)");
  return ROVIT_OK;
}

// C-ABI form of rovit_wgrad_batch: up to 4 weight gradients that share M in one launch (the per-block launch of
// rovit_vit_backward); arrays of n entries.  ws[j] as for rovit_wgrad with the same `splits`.
extern "C" int rovit_wgrad_multi(const void* const* dY, const int* ldy, const void* const* A, const int* lda, const int* N, const int* K,
                                 float* const* ws, int n, int M, int splits, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && ldy && A && lda && N && K && ws && n >= 1 && n <= WG_MAXPROB, ROVIT_ERR_NULL, "wgrad_multi: bad arguments");
  RovitWgradDesc d[WG_MAXPROB];
  for (int j = 0; j < n; ++j) d[j] = {dY[j], ldy[j], A[j], lda[j], N[j], K[j], ws[j]};
  return rovit_wgrad_batch(d, n, M, splits, stream);
}

// rovit_wgrad_multi with per-problem operand layouts: a_blk[j] / y_blk[j] != 0 = A / dY of problem j is stored chunk-major
// [cols / 32][M][32] (what rovit_mlp_fused_fwd / _bwd write: act and dpre); NULL arrays = all row-major
extern "C" int rovit_wgrad_multi_ex(const void* const* dY, const int* ldy, const void* const* A, const int* lda, const int* N, const int* K,
                                    float* const* ws, const int* a_blk, const int* y_blk, int n, int M, int splits, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && ldy && A && lda && N && K && ws && n >= 1 && n <= WG_MAXPROB, ROVIT_ERR_NULL, "wgrad_multi_ex: bad arguments");
  RovitWgradDesc d[WG_MAXPROB];
  for (int j = 0; j < n; ++j) d[j] = {dY[j], ldy[j], A[j], lda[j], N[j], K[j], ws[j], a_blk ? a_blk[j] : 0, y_blk ? y_blk[j] : 0};
  return rovit_wgrad_batch(d, n, M, splits, stream);
}

extern "C" int rovit_wgrad_reduce(const float* ws, int splits, int N, int K, const float* gamma, const float* beta,
                                  const float* W, float* dW, float* db, float* dgamma, float* dbeta, float* g_scratch,
                                  rovit_stream_t stream) {
  const RovitReduceDesc d{ws, splits, N, K, gamma, beta, W, dW, db, dgamma, dbeta, g_scratch};
  return rovit_wgrad_reduce_batch(&d, 1, stream);
}

// reduce (and un-fold) up to ROVIT_REDUCE_BATCH weight gradients in two launches
int rovit_wgrad_reduce_batch(const RovitReduceDesc* descs, int n, rovit_stream_t stream) {
#ifdef ROVIT_DEV
  if (ROVIT_KNOB(ROVIT_KNOB_SKIP_WGRAD_REDUCE, 0)) return ROVIT_OK;      // timing experiment only: gradients are wrong
#endif
  ROVIT_CHECK_ARG(descs && n > 0 && n <= ROVIT_REDUCE_BATCH, ROVIT_ERR_SHAPE, "wgrad_reduce_batch: bad batch size %d", n);
  ReduceBatch rb{}, ab{};
  int blocks = 0, fblocks = 0;
  for (int i = 0; i < n; ++i) {
    const RovitReduceDesc& d = descs[i];
    ROVIT_CHECK_ARG(d.ws && d.dW && d.db, ROVIT_ERR_NULL, "wgrad_reduce_batch: null pointer");
    rb.d[i] = d;
    rb.first_block[i] = blocks;
    if (d.gamma) {
      ROVIT_CHECK_ARG(d.beta && d.W && d.dgamma && d.dbeta && d.g_scratch, ROVIT_ERR_NULL, "wgrad_reduce_batch: affine un-fold needs beta/W/outputs");
      ROVIT_CHECK_ARG(d.N % 16 == 0 && d.K % 64 == 0 && 2 * (d.N / 16) <= d.N, ROVIT_ERR_SHAPE, "wgrad_reduce_batch: affine un-fold needs N %% 16 == 0, K %% 64 == 0");
      blocks += (d.N / 16) * (d.K / 64);
      ab.d[ab.n] = d;
      ab.first_block[ab.n] = fblocks;
      fblocks += (d.K + 255) / 256;
      ab.n++;
    } else {
      const int nthreads = d.N * d.K / 4 > d.N ? d.N * d.K / 4 : d.N;
      blocks += (nthreads + 255) / 256;
    }
  }
  rb.n = n; rb.first_block[n] = blocks;
  ab.first_block[ab.n] = fblocks;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rb);
  ROVIT_CHECK_LAUNCH("wgrad_reduce_batch_kernel");
  if (ab.n > 0) {
    hipLaunchKernelGGL(wgrad_affine_finalize_kernel, dim3(fblocks), dim3(256), 0, (hipStream_t)stream, ab);
    ROVIT_CHECK_LAUNCH("wgrad_affine_finalize_kernel");
  }
  return ROVIT_OK;
}

// X(M,192) += A(M,K) W(192,K)^T + bias (branch output rounded to bf16 first), fused with the LayerNorm that
// follows the residual add: xhat_out (bf16) / rstd_out of the updated rows (pass NULL for no LayerNorm).
extern "C" int rovit_gemm_resid_ln(const void* A, int lda, const void* W, int ldw, int M, int K, const float* bias, float* X,
                                   void* xhat_out, float* rstd_out, float eps, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(A && W && X, ROVIT_ERR_NULL, "gemm_resid_ln: null pointer");
  ROVIT_CHECK_ARG(M > 0 && (K == 192 || K == 576 || K == 768), ROVIT_ERR_SHAPE, "gemm_resid_ln: K must be 192/576/768 (got %d)", K);
  ROVIT_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && rovit_aligned16(A) && rovit_aligned16(W) && rovit_aligned16(X), ROVIT_ERR_ALIGN,
                  "gemm_resid_ln: alignment");
  ROVIT_CHECK_ARG(!xhat_out || rstd_out, ROVIT_ERR_NULL, "gemm_resid_ln: rstd_out missing");
  GemmArgs g{};
  g.A = (const bf16*)A; g.lda = lda; g.W = (const bf16*)W; g.ldw = ldw; g.M = M; g.N = 192; g.K = K; g.bias = bias;
  g.xres = X; g.ldx = 192; g.out = (bf16*)xhat_out; g.ldo = 192; g.rstd_out = rstd_out; g.eps = eps; GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0));
  if (K == 192) return launch_ws<6, 1, 64>(g, EPI_RESID_LN, (hipStream_t)stream);
  if (kdma_enabled() && xhat_out && K == 576) return launch_kdma<18, EPI_RESID_LN>(g, (hipStream_t)stream);
  if (K == 576) return launch_ws<9, 2, 32>(g, EPI_RESID_LN, (hipStream_t)stream);
  return launch_ws<12, 2, 32>(g, EPI_RESID_LN, (hipStream_t)stream);
}

// dgrad + LayerNorm backward: dxhat = dY(M,K) W(192,K)^T, then dX += rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)),
// dXb = bf16(dX).  (W is the transposed folded weight, so the LayerNorm affine is already applied.)
// dXb_in (round 4, may be NULL): the incoming residual gradient as bf16 rows (M,192).  Given, the launch computes
// dXb = bf16(float(dXb_in) + LayerNorm-backward(dxhat)) and neither reads nor writes the fp32 dX (which may then be NULL): the residual
// gradient travels between the kernels of rovit_vit_backward in bf16 (58 MB per launch less at batch 256), summed in fp32 inside each.
static int gemm_ln_bwd_impl(const void* dY, int ldy, const void* W, int ldw, int M, int K, const void* xhat, const float* rstd,
                            float* dX, const void* dXb_in, int cls_step, void* dXb, rovit_stream_t stream);
extern "C" int rovit_gemm_ln_bwd(const void* dY, int ldy, const void* W, int ldw, int M, int K, const void* xhat, const float* rstd,
                                 float* dX, const void* dXb_in, void* dXb, rovit_stream_t stream) {
  return gemm_ln_bwd_impl(dY, ldy, W, ldw, M, K, xhat, rstd, dX, dXb_in, 0, dXb, stream);
}
int rovit_gemm_ln_bwd_cls(const void* dY, int ldy, const void* W, int ldw, int M, int K, const void* xhat, const float* rstd,
                          const void* dXb_in, int cls_step, void* dXb, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dXb_in && cls_step > 0, ROVIT_ERR_NULL, "gemm_ln_bwd_cls: needs the bf16 incoming gradient and a row step");
  return gemm_ln_bwd_impl(dY, ldy, W, ldw, M, K, xhat, rstd, nullptr, dXb_in, cls_step, dXb, stream);
}
static int gemm_ln_bwd_impl(const void* dY, int ldy, const void* W, int ldw, int M, int K, const void* xhat, const float* rstd,
                            float* dX, const void* dXb_in, int cls_step, void* dXb, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && W && xhat && rstd && (dX || dXb_in) && dXb, ROVIT_ERR_NULL, "gemm_ln_bwd: null pointer");
  ROVIT_CHECK_ARG(rovit_aligned16(dXb_in) && rovit_aligned16(dXb), ROVIT_ERR_ALIGN, "gemm_ln_bwd: alignment");
  ROVIT_CHECK_ARG(M > 0 && (K == 192 || K == 576 || K == 768), ROVIT_ERR_SHAPE, "gemm_ln_bwd: K must be 192/576/768 (got %d)", K);
  ROVIT_CHECK_ARG(ldy % 8 == 0 && ldw % 8 == 0 && rovit_aligned16(dY) && rovit_aligned16(W) && rovit_aligned16(dX), ROVIT_ERR_ALIGN,
                  "gemm_ln_bwd: alignment");
  GemmArgs g{};
  g.A = (const bf16*)dY; g.lda = ldy; g.W = (const bf16*)W; g.ldw = ldw; g.M = M; g.N = 192; g.K = K;
  g.mul = (const bf16*)xhat; g.ldm = 192; g.pos = rstd; g.xres = dX; g.ldx = 192; g.out = (bf16*)dXb; g.ldo = 192; g.xprev = (const bf16*)dXb_in; g.tokens = cls_step; GEMM_SET_DBG(g, ROVIT_KNOB(ROVIT_KNOB_GEMM_DBG, 0) | (ROVIT_KNOB(ROVIT_KNOB_SKIP_DX_FP32_STORE, 0) ? 64 : 0));
  if (K == 192) return launch_ws<6, 1, 64>(g, EPI_LNBWD, (hipStream_t)stream);
  if (kdma_enabled() && K == 576) return launch_kdma<18, EPI_LNBWD>(g, (hipStream_t)stream);
  if (K == 576) return launch_ws<9, 2, 32>(g, EPI_LNBWD, (hipStream_t)stream);
  return launch_ws<12, 2, 32>(g, EPI_LNBWD, (hipStream_t)stream);
}
