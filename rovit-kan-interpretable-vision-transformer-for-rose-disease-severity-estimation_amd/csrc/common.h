// Shared device/host helpers for the gfx950 kernels of the RoViT-KAN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rovit_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define ROVIT_WAVE 64

// ---- error plumbing ------------------------------------------------------------------
void rovit_set_error(const char* fmt, ...);

#define ROVIT_CHECK_ARG(cond, code, ...)        \
  do {                                          \
    if (!(cond)) {                              \
      rovit_set_error(__VA_ARGS__);             \
      return (code);                            \
    }                                           \
  } while (0)

#define ROVIT_CHECK_LAUNCH(name)                                                  \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      rovit_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return ROVIT_ERR_LAUNCH;                                                    \
    }                                                                             \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: (kernel, device) pairs are remembered,
// not a process-wide flag, so a second device in the same process gets its attribute too (api.hip)
bool rovit_set_max_lds(const void* fn, size_t bytes);

static inline bool rovit_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// ---- developer knobs (A/B timing, ablations) ------------------------------------------------------------
// The PRODUCT library has none: ROVIT_KNOB(id, dflt) is the constant `dflt`, no debug entry point is exported, no kernel
// that the default path cannot reach is compiled.  `make dev` builds lib/librovit_hip_dev.so with -DROVIT_DEV: there
// rovit_dev_set_knob(id, value) (api.hip) overrides a knob for the calls that follow -- used by tools/ only
// (ROVIT_HIP_LIB=.../librovit_hip_dev.so).
enum RovitKnob {
  ROVIT_KNOB_ATTN_FWD_R3 = 0,   // 1: round 3's attention forward (one workgroup per CU)
  ROVIT_KNOB_ATTN_BWD_R3 = 1,   // 1: round 3's attention backward (one workgroup per CU, four LDS tiles)
  ROVIT_KNOB_ATTN_DBG = 2,      // attention backward ablation bits (skip pass 1 / pass 2; results are then wrong)
  ROVIT_KNOB_ATTN_BWD_SPLIT = 3, // 1: attention backward with the two passes as separate workgroups (measured slower: 62 us against 50)
  ROVIT_KNOB_SINGLE_STREAM = 4,  // 1: forward and backward entirely on the caller's stream (serial per-kernel times for the profiles)
  ROVIT_KNOB_WGRAD_SPLITS = 5,   // M-splits of the merged weight-gradient launch (default 16)
  ROVIT_KNOB_GEMM_DBG = 6,       // GEMM / weight-gradient ablation bits (skip stores / MFMAs / loads; results are then wrong)
  ROVIT_KNOB_MLP_DBG = 7,        // fused-MLP ablation bits (skip epilogue / GELU / fc2 / fc1; results are then wrong)
  ROVIT_KNOB_MLP_SCHEDULE = 8,   // fused MLP forward: 10 (default) in-wave pipeline + GELU table, 8 lockstep, 9 staggered, 4 two 128-row workgroups
  ROVIT_KNOB_MLP_RPW = 9,        // token rows per workgroup of the 8-wave MLP kernels (default: by size, 240 or 256)
  ROVIT_KNOB_WGRAD_TILE = 10,    // (tn << 16) | tk: output tile of the single-problem weight-gradient launch
  ROVIT_KNOB_WGRAD_WGS = 11,     // workgroups the single-problem weight-gradient launch aims for (default 512)
  ROVIT_KNOB_KAN_MFMA_NS = 12,   // sample tiles per wave of the matrix-core KAN stack (default: by grid size)
  ROVIT_KNOB_KDMA = 13,          // 0: the register-staged kernels instead of gemm_kdma_kernel for the K = 576 dgrad
  ROVIT_KNOB_SKIP_DX_FP32_STORE = 15, // 1: the LayerNorm-backward epilogues do not write the fp32 dX (gradients WRONG): bound of a bf16 residual-gradient stream
  ROVIT_KNOB_SIDE_PRIORITY = 16, // priority class of the library's side stream: 0 highest (default), 1 lowest, 2 a normal-priority stream
  ROVIT_KNOB_PROJ_DGRAD_CUS = 17, // CUs the proj dgrad launch of the backward sizes its grid for (default 256)
  ROVIT_KNOB_TAIL_WAVES = 18,    // waves per workgroup of the forward block tail: 8 (two row tiles per wave) or 16 (one; four waves per SIMD)
  ROVIT_KNOB_FWD_STAGGER = 19,   // 1: the forward's second half-batch starts one attention launch behind the first
  ROVIT_KNOB_ATTN_FWD3 = 20,     // 1: attention forward with three workgroups per CU (attn_fwd3_kernel)
  ROVIT_KNOB_SKIP_WGRAD_REDUCE = 14,  // 1: the slab-reduce / affine-finalize launches are not issued (gradients WRONG): upper bound of what folding them away could gain
  ROVIT_KNOB_COUNT = 32
};
#ifdef ROVIT_DEV
extern int g_rovit_knob[ROVIT_KNOB_COUNT];
extern bool g_rovit_knob_set[ROVIT_KNOB_COUNT];
#define ROVIT_KNOB(id, dflt) (g_rovit_knob_set[id] ? g_rovit_knob[id] : (dflt))
#else
#define ROVIT_KNOB(id, dflt) (dflt)
#endif

// ---- device helpers ------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  // D[row][col] += sum_k A[row][k] B[k][col];  lane l supplies A[row = l&15][k-slots of group l>>4]
  // and B[k-slots][col = l&15]; holds D[row = 4*(l>>4)+r][col = l&15] in c[r].
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns
// 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i (row q in element q).
__device__ __forceinline__ bf16x4 lds_read_tr(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(p));
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (bf16)lo[0]; r[1] = (bf16)lo[1]; r[2] = (bf16)lo[2]; r[3] = (bf16)lo[3];
  r[4] = (bf16)hi[0]; r[5] = (bf16)hi[1]; r[6] = (bf16)hi[2]; r[7] = (bf16)hi[3];
  return r;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// v if keep else zeros, as four dword selects (no branch, so the producing load is not waited for early)
__device__ __forceinline__ bf16x8 keep_if(bf16x8 v, bool keep) {
  u32x4 u = __builtin_bit_cast(u32x4, v);
  u[0] = keep ? u[0] : 0u; u[1] = keep ? u[1] : 0u; u[2] = keep ? u[2] : 0u; u[3] = keep ? u[3] : 0u;
  return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ bf16x4 pack4(f32x4 v) {
  bf16x4 r;
  r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
  return r;
}

// Workgroup ids b, b+8, b+16, ... are observed to share an XCD (round-robin dispatch); this bijection gives
// XCD x the contiguous id range [start_x, start_x + count_x) so neighbouring tiles share one L2.
// Speed only: nothing depends on the placement actually happening.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt(0), which
// would serialise every prefetched global load behind the barrier; plain loads are still waited for by the
// compiler at their first use.
__device__ __forceinline__ void barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum16(float v) {   // sum over the 16 lanes sharing l>>4
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}
// same sum with DPP row rotations (no LDS-pipeline instruction): every lane of a 16-lane row gets the row total
__device__ __forceinline__ float row_sum16_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
  v = wave_sum16(v); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}
// reduce over the 4 lane groups (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float group4_sum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
__device__ __forceinline__ float group4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// GELU (exact-erf form) and its derivative from ONE exponential: with z = |x|/sqrt(2), e = exp(-z^2) = exp(-x^2/2),
// erf(z) = 1 - (a1 t + ... + a5 t^5) e, t = 1/(1 + p z)   (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7),
// Phi(x) = 0.5 (1 + sign(x) erf(z)),  gelu = x Phi,  gelu' = Phi + x e / sqrt(2 pi).
// The reciprocal is the hardware v_rcp_f32 (1 ulp; an IEEE division costs ~10 instructions here and this runs on
// 38.7 M elements per layer), the exponential one v_exp_f32 on a pre-scaled argument, and the 0.5 of Phi is folded
// into the polynomial coefficients: h = 0.5 erfc(z) = (a1/2 t + ...) e, Phi = 0.5 + sign(x) (0.5 - h).
__device__ __forceinline__ void gelu_and_grad(float x, float& act, float& dact) {
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);          // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189303853130f, fabsf(x), 1.f));   // 1/(1 + p |x|/sqrt 2)
  float p = fmaf(0.5306027145f, t, -0.7265760135f);
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  const float half_erf = fmaf(-p * t, e, 0.5f);                                     // 0.5 erf(|x|/sqrt 2)
  const float cdf = 0.5f + copysignf(half_erf, x);
  act = x * cdf;
  dact = fmaf(x * 0.3989422804014327f, e, cdf);
}

// the derivative alone (same formulas, one multiply less)
__device__ __forceinline__ float gelu_grad(float x) {
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189303853130f, fabsf(x), 1.f));
  float p = fmaf(0.5306027145f, t, -0.7265760135f);
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  const float cdf = 0.5f + copysignf(fmaf(-p * t, e, 0.5f), x);
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

#endif  // __HIPCC__

// ---- internal batched helpers (not part of the C ABI): many small problems in one launch, descriptors passed
// by value in the kernel arguments ------------------------------------------------------------------------
struct RovitPrepDesc {
  const float* W; const float* bias; const float* gamma; const float* beta;
  void* Wf; void* WfT; float* bias_f;
  int N, K;
};
constexpr int ROVIT_PREP_BATCH = 52;     // 1 + 4*12 descriptors of a depth-12 backbone in ONE launch (3.5 KB of kernel arguments)
int rovit_prep_weight_batch(const RovitPrepDesc* descs, int n, rovit_stream_t stream);

// LayerNorm forward on every row_step-th row of the dense (rows*row_step, 192) buffers, in place (the taps path of the last block)
void rovit_set_cu_budget(int cus);   // gemm.hip: CUs the weight-stationary GEMM launches size their grids for
// vit.hip: the device's second stream (created on first use; nullptr on failure or when the developer library asks for a single stream)
hipStream_t rovit_side_stream_handle();
int rovit_layernorm_fwd_rows(const float* x, void* xhat, float* rstd, int rows, int row_step, float eps, rovit_stream_t stream);
int rovit_cls_norm_affine_grad(const float* dfeat, const float* xhat, float* dgamma, float* dbeta, int batch, rovit_stream_t stream);
// cls_tail.hip: the last block's post-attention half + the final norm on the class-token rows in one launch
int rovit_cls_tail_fwd(const void* o, float* X, const void* wproj, const float* bproj, const void* wfc1, const float* bfc1, const void* wfc2,
                       const float* bfc2, const float* gamma, const float* beta, void* xhat2, float* rstd2, void* act, void* dact, float* feat,
                       float* xhat_cls, float* rstd_cls, int rows, int tokens, float eps, rovit_stream_t stream);
int rovit_cls_tail_bwd(const float* dfeat, const float* xhat_cls, const float* rstd_cls, const float* gamma, const void* wfc2, const void* wfc1,
                       const void* wproj, const void* dact, const void* xhat2, const float* rstd2, void* xin, void* dpre, void* xmid, void* dO,
                       int rows, int tokens, rovit_stream_t stream);
// attention backward whose dout carries gradient on the first `dout_rows` rows of every image only (the last block: the class token's)
int rovit_attention_bwd_rows(const void* qkv, const void* out, const float* lse2, const void* dout, int dout_rows, void* dqkv, int batch,
                             int tokens, int heads, int head_dim, float scale, rovit_stream_t stream);
// rovit_gemm_ln_bwd with a bf16 residual gradient that is non-zero on every `cls_step`-th row only (other rows: zero, not read)
int rovit_gemm_ln_bwd_cls(const void* dY, int ldy, const void* W, int ldw, int M, int K, const void* xhat, const float* rstd,
                          const void* dXb_in, int cls_step, void* dXb, rovit_stream_t stream);
// several weight gradients G[N][K] = dY[M][N]^T A[M][K] that share M, in one launch (gemm.hip)
struct RovitWgradDesc { const void* dY; int ldy; const void* A; int lda; int N, K; float* ws; int a_blk, y_blk; };   // *_blk: operand chunk-major [cols/32][M][32]
int rovit_wgrad_batch(const RovitWgradDesc* descs, int n, int M, int splits, rovit_stream_t stream);

struct RovitReduceDesc {
  const float* ws; int splits, N, K;
  const float* gamma; const float* beta; const float* W;      // gamma != NULL: un-fold the LayerNorm affine
  float* dW; float* db; float* dgamma; float* dbeta; float* g_scratch;
};
// mlp_fused.hip: weight streams of `depth` blocks laid out inside the prepared-weight buffer (byte offsets)
int rovit_mlp_stream_prep_blocks(const void* prep_base, size_t blk0, size_t stride, size_t off_w1, size_t off_w2, size_t off_out,
                                 size_t off_wp, int bwd, size_t off_wq_next, int depth, rovit_stream_t stream, bool gelu_tables = true);
constexpr int ROVIT_REDUCE_BATCH = 4;
int rovit_wgrad_reduce_batch(const RovitReduceDesc* descs, int n, rovit_stream_t stream);
