// fp32 reference-precision forward of the DeiT-Tiny backbone (inference only).
//
// Why it exists: BASELINE.json's north_star asks for logits/severity within 1e-3 of the reference's CPU path in fp32 and
// a bit-exact class argmax.  The training path computes its GEMMs with bf16 operands (the precision class of the
// reference's own CUDA autocast path), which moves the features by ~5e-3 RMS -- enough to carry a feature across the
// discontinuity of the reference's truncated spline.  This path runs the SAME arithmetic (timm VisionTransformer.forward,
// SURVEY.md section 2, reached from /root/reference/models/backbone.py:12-25) entirely in fp32 on the GPU, so the end-to-end
// parity statement can be made at fp32 tolerance.  It is a parity / evaluation mode, not the fast path: plain tiled VALU
// kernels (exact fp32 FMAs, no MFMA, no bf16 anywhere), ~30x slower than the bf16 path.
#include "common.h"

namespace {

constexpr int T = 197, D = 192, H = 3, HD = 64, MLP = 768, PD = 768;
enum { P_CLS = 0, P_POS, P_PATCH_W, P_PATCH_B, P_NORM_W, P_NORM_B, P_BLOCK0 };
enum { B_N1W = 0, B_N1B, B_QKVW, B_QKVB, B_PROJW, B_PROJB, B_N2W, B_N2B, B_FC1W, B_FC1B, B_FC2W, B_FC2B, B_COUNT };
enum { F_NONE = 0, F_GELU = 1, F_RESID = 2, F_PATCH = 3 };

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// C[M,N] = A[M,K] W[N,K]^T + bias, 64 x 64 tile, 256 threads, thread = 4 x 4 outputs, K in steps of 16 through LDS.
//   F_GELU : exact-erf GELU on the result          F_RESID: C += result (in place on the residual stream)
//   F_PATCH: A is gathered from the NCHW image (row m = image b, patch p; k = c*256 + py*16 + px), the result goes to token
//            row b*T + 1 + p with the position embedding added (timm PatchEmbed + pos_embed)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N, int K,
                                                       const float* __restrict__ pos) {
  __shared__ float As[16][64 + 4], Ws[16][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    // stage A (64 x 16) and W (64 x 16), transposed to [k][row]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * 256, r = e >> 4, kk = e & 15;
      const int m = m0 + r, n = n0 + r, k = k0 + kk;
      float av = 0.f;
      if (m < M) {
        if (EPI == F_PATCH) {
          const int b = m / (T - 1), p = m - b * (T - 1);
          const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
          av = A[(((size_t)b * 3 + c) * 224 + (p / 14) * 16 + py) * 224 + (p % 14) * 16 + px];
        } else {
          av = A[(size_t)m * lda + k];
        }
      }
      As[kk][r] = av;
      Ws[kk][r] = n < N ? W[(size_t)n * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 a4 = *(const float4*)&As[kk][ty * 4];
      const float4 w4 = *(const float4*)&Ws[kk][tx * 4];
      const float a[4] = {a4.x, a4.y, a4.z, a4.w}, w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j] + (bias ? bias[n] : 0.f);
      if (EPI == F_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));      // exact-erf GELU (timm default)
      if (EPI == F_PATCH) {
        const int b = m / (T - 1), p = m - b * (T - 1);
        C[((size_t)b * T + 1 + p) * ldc + n] = v + pos[(size_t)(1 + p) * N + n];
      } else if (EPI == F_RESID) {
        C[(size_t)m * ldc + n] += v;
      } else {
        C[(size_t)m * ldc + n] = v;
      }
    }
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

// softmax(q k^T * scale) v for one (image, head): K and V (197 x 64 fp32) in LDS, one thread per query row, online softmax
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + T * HD;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int ld = 3 * D;
  const float* base = qkv + (size_t)b * T * ld + h * HD;
  for (int e = threadIdx.x; e < T * HD; e += 256) {
    const int r = e >> 6, c = e & 63;
    Ks[e] = base[(size_t)r * ld + D + c];
    Vs[e] = base[(size_t)r * ld + 2 * D + c];
  }
  __syncthreads();
  const int qr = threadIdx.x;
  if (qr >= T) return;
  float q[HD], o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = base[(size_t)qr * ld + d] * scale; o[d] = 0.f; }      // timm scales q before the product
  float m = -INFINITY, l = 0.f;
  for (int key = 0; key < T; ++key) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(q[d], Ks[key * HD + d], s);
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn), p = expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = fmaf(p, Vs[key * HD + d], o[d] * corr);
    m = mn;
  }
  const float inv = 1.f / l;
  float* dst = out + ((size_t)b * T + qr) * D + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) dst[d] = o[d] * inv;
}

__global__ __launch_bounds__(256) void cls_rows_f32_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ X, int B) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * D) return;
  const int b = e / D, c = e - b * D;
  X[(size_t)b * T * D + c] = cls[c] + pos[c];
}

template <int EPI>
int gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, const float* pos, hipStream_t st) {
  hipLaunchKernelGGL((gemm_f32_kernel<EPI>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, lda, W, bias, C, ldc, M, N, K, pos);
  ROVIT_CHECK_LAUNCH("gemm_f32_kernel");
  return ROVIT_OK;
}

#define RUN(call) do { int rc__ = (call); if (rc__ != ROVIT_OK) return rc__; } while (0)

}  // namespace

// workspace: X (M,192) + xn (M,192) + qkv (M,576) + o (M,192) + h (M,768), all fp32, M = batch * 197
extern "C" size_t rovit_vit_f32_workspace_bytes(int batch) {
  const size_t M = (size_t)batch * T;
  return al(M * D * 4) * 3 + al(M * 3 * D * 4) + al(M * MLP * 4);
}

// images fp32 NCHW (B,3,224,224) -> features fp32 (B,192), every operation in fp32 (see the file header).  params as for
// rovit_vit_forward (the ORIGINAL fp32 parameters; no prepared weights).
extern "C" int rovit_vit_forward_f32(const float* images, const float* const* params, void* workspace, float* features, int batch, int depth,
                                     rovit_stream_t stream) {
  ROVIT_CHECK_ARG(images && params && workspace && features, ROVIT_ERR_NULL, "vit_forward_f32: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && depth > 0, ROVIT_ERR_SHAPE, "vit_forward_f32: bad batch/depth");
  hipStream_t st = (hipStream_t)stream;
  const int M = batch * T;
  char* ws = (char*)workspace;
  size_t o = 0;
  float* X = (float*)(ws + o); o += al((size_t)M * D * 4);
  float* xn = (float*)(ws + o); o += al((size_t)M * D * 4);
  float* ao = (float*)(ws + o); o += al((size_t)M * D * 4);
  float* qkv = (float*)(ws + o); o += al((size_t)M * 3 * D * 4);
  float* hbuf = (float*)(ws + o);
  const float eps = 1e-6f;
  hipLaunchKernelGGL(cls_rows_f32_kernel, dim3((batch * D + 255) / 256), dim3(256), 0, st, params[P_CLS], params[P_POS], X, batch);
  ROVIT_CHECK_LAUNCH("cls_rows_f32_kernel");
  RUN(gemm_f32<F_PATCH>(images, 0, params[P_PATCH_W], params[P_PATCH_B], X, D, batch * (T - 1), D, PD, params[P_POS], st));
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_f32_kernel, (size_t)2 * T * HD * 4), ROVIT_ERR_LAUNCH, "vit_forward_f32: cannot raise the LDS limit");
  for (int i = 0; i < depth; ++i) {
    const float* const* bp = params + P_BLOCK0 + B_COUNT * i;
    hipLaunchKernelGGL(ln_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, st, X, bp[B_N1W], bp[B_N1B], xn, M, 1, eps);
    ROVIT_CHECK_LAUNCH("ln_f32_kernel");
    RUN(gemm_f32<F_NONE>(xn, D, bp[B_QKVW], bp[B_QKVB], qkv, 3 * D, M, 3 * D, D, nullptr, st));
    hipLaunchKernelGGL(attn_f32_kernel, dim3(batch * H), dim3(256), (size_t)2 * T * HD * 4, st, qkv, ao, 0.125f);
    ROVIT_CHECK_LAUNCH("attn_f32_kernel");
    RUN(gemm_f32<F_RESID>(ao, D, bp[B_PROJW], bp[B_PROJB], X, D, M, D, D, nullptr, st));
    hipLaunchKernelGGL(ln_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, st, X, bp[B_N2W], bp[B_N2B], xn, M, 1, eps);
    ROVIT_CHECK_LAUNCH("ln_f32_kernel");
    RUN(gemm_f32<F_GELU>(xn, D, bp[B_FC1W], bp[B_FC1B], hbuf, MLP, M, MLP, D, nullptr, st));
    RUN(gemm_f32<F_RESID>(hbuf, MLP, bp[B_FC2W], bp[B_FC2B], X, D, M, D, MLP, nullptr, st));
  }
  // final LayerNorm on the class token of every image (row step T)
  hipLaunchKernelGGL(ln_f32_kernel, dim3((batch + 3) / 4), dim3(256), 0, st, X, params[P_NORM_W], params[P_NORM_B], features, batch, T, eps);
  ROVIT_CHECK_LAUNCH("ln_f32_kernel");
  return ROVIT_OK;
}
