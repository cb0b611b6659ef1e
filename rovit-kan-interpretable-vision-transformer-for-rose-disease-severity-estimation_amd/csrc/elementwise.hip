// HBM-bound row kernels of the DeiT-Tiny path: LayerNorm forward/backward on the fp32 residual stream,
// patch im2col, cls/pos handling, final-norm on the CLS rows, and the per-step weight preparation
// (LayerNorm affine folded into the following Linear, bf16 cast, transposed copies for dgrad).
//
// Reference arithmetic being restated: timm VisionTransformer (LayerNorm eps=1e-6, PatchEmbed conv k16/s16,
// cls_token / pos_embed, final norm + token 0) reached through /root/reference/models/backbone.py:12-25.
#include "common.h"

namespace {

constexpr int D = 192;          // embed dim; one row = 16 lanes x 3 float4

// ---- LayerNorm forward: x fp32 (M,192) -> xhat bf16 (M,192), rstd (M) ------------------------------------
// 16 lanes per row; lane c holds elements {64*i + 4*c .. +3 : i = 0..2}: every load/store instruction of a
// 16-lane group covers 256 (fp32) / 128 (bf16) contiguous bytes.
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, bf16* __restrict__ xhat,
                                                     float* __restrict__ rstd, int M, float eps, size_t x_ld, size_t h_ld,
                                                     size_t r_ld) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  if (row >= M) return;
  const float4* xr = (const float4*)(x + (size_t)row * x_ld);
  float4 v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { v[i] = xr[16 * i + c]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
  const float mean = wave_sum16(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  }
  const float r = rsqrtf(wave_sum16(q) * (1.f / D) + eps);
  bf16x4* o = (bf16x4*)(xhat + (size_t)row * h_ld);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f32x4 t = {v[i].x * r, v[i].y * r, v[i].z * r, v[i].w * r};
    o[16 * i + c] = pack4(t);
  }
  if (c == 0) rstd[(size_t)row * r_ld] = r;
}

// ---- LayerNorm backward (affine already folded into dxhat by the dgrad GEMM) -----------------------------
// dX += rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat));  dXb = bf16(dX)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dxhat, const bf16* __restrict__ xhat,
                                                     const float* __restrict__ rstd, float* __restrict__ dX,
                                                     bf16* __restrict__ dXb, int M, size_t ld, size_t r_ld) {
  // every row-major operand uses the same row stride `ld` (D when dense, T*D for the CLS rows only)
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  if (row >= M) return;
  const bf16x4* gp = (const bf16x4*)(dxhat + (size_t)row * ld);
  const bf16x4* hp = (const bf16x4*)(xhat + (size_t)row * ld);
  float g[12], h[12];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bf16x4 a = gp[16 * i + c], b = hp[16 * i + c];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[4 * i + e] = (float)a[e]; h[4 * i + e] = (float)b[e];
      s1 += g[4 * i + e]; s2 += g[4 * i + e] * h[4 * i + e];
    }
  }
  const float c1 = wave_sum16(s1) * (1.f / D), c2 = wave_sum16(s2) * (1.f / D);
  const float r = rstd[(size_t)row * r_ld];
  float4* xp = (float4*)(dX + (size_t)row * ld);
  bf16x4* bp = (bf16x4*)(dXb + (size_t)row * ld);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float4 v = xp[16 * i + c];
    v.x += r * (g[4 * i + 0] - c1 - h[4 * i + 0] * c2);
    v.y += r * (g[4 * i + 1] - c1 - h[4 * i + 1] * c2);
    v.z += r * (g[4 * i + 2] - c1 - h[4 * i + 2] * c2);
    v.w += r * (g[4 * i + 3] - c1 - h[4 * i + 3] * c2);
    xp[16 * i + c] = v;
    f32x4 t = {v.x, v.y, v.z, v.w};
    bp[16 * i + c] = pack4(t);
  }
}

// ---- patch im2col: x fp32 NCHW (B,3,224,224) -> (B*196, 768) bf16, column = c*256 + kh*16 + kw ----------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, bf16* __restrict__ col, int B) {
  // one thread = 8 consecutive kw of one (patch, c, kh)
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)B * 196 * 96;
  if (e >= total) return;
  const int chunk = (int)(e % 96);
  const size_t patch = e / 96;
  const int b = (int)(patch / 196), p = (int)(patch - (size_t)b * 196);
  const int ph = p / 14, pw = p - ph * 14;
  const int c = chunk / 32, rem = chunk - c * 32, kh = rem >> 1, half = rem & 1;
  const float* src = x + (((size_t)b * 3 + c) * 224 + ph * 16 + kh) * 224 + pw * 16 + half * 8;
  const float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
  f32x4 lo = {a.x, a.y, a.z, a.w}, hi = {d.x, d.y, d.z, d.w};
  *(bf16x8*)(col + patch * 768 + chunk * 8) = pack8(lo, hi);
}

// X[b*T + 0][:] = cls + pos[0]
__global__ __launch_bounds__(256) void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                      float* __restrict__ X, int B, int T) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * D) return;
  const int b = e / D, d = e - b * D;
  X[(size_t)b * T * D + d] = cls[d] + pos[d];
}

// ---- final norm on the CLS rows (fp32, explicit affine): features = LN(X[b*T]) -----------------------------
__global__ __launch_bounds__(256) void cls_ln_fwd_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ feat,
                                                         float* __restrict__ xhat_out, float* __restrict__ rstd_out,
                                                         int B, int T, float eps) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  if (b >= B) return;
  const float4* xr = (const float4*)(X + (size_t)b * T * D);
  float4 v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { v[i] = xr[16 * i + c]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
  const float mean = wave_sum16(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  }
  const float r = rsqrtf(wave_sum16(q) * (1.f / D) + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 g = ((const float4*)gamma)[16 * i + c], bt = ((const float4*)beta)[16 * i + c];
    const float4 h = make_float4(v[i].x * r, v[i].y * r, v[i].z * r, v[i].w * r);
    ((float4*)(feat + (size_t)b * D))[16 * i + c] = make_float4(h.x * g.x + bt.x, h.y * g.y + bt.y, h.z * g.z + bt.z, h.w * g.w + bt.w);
    if (xhat_out) ((float4*)(xhat_out + (size_t)b * D))[16 * i + c] = h;
  }
  if (c == 0 && rstd_out) rstd_out[b] = r;
}

// dX[b*T][:] = LN backward of dfeat (all other rows of dX/dXb must already be zero)
__global__ __launch_bounds__(256) void cls_ln_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         float* __restrict__ dX, bf16* __restrict__ dXb, int B, int T) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  if (b >= B) return;
  float g[12], h[12];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 df = ((const float4*)(dfeat + (size_t)b * D))[16 * i + c];
    const float4 gm = ((const float4*)gamma)[16 * i + c];
    const float4 xh = ((const float4*)(xhat + (size_t)b * D))[16 * i + c];
    g[4 * i + 0] = df.x * gm.x; g[4 * i + 1] = df.y * gm.y; g[4 * i + 2] = df.z * gm.z; g[4 * i + 3] = df.w * gm.w;
    h[4 * i + 0] = xh.x; h[4 * i + 1] = xh.y; h[4 * i + 2] = xh.z; h[4 * i + 3] = xh.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1 += g[4 * i + e]; s2 += g[4 * i + e] * h[4 * i + e]; }
  }
  const float c1 = wave_sum16(s1) * (1.f / D), c2 = wave_sum16(s2) * (1.f / D);
  const float r = rstd[b];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = r * (g[4 * i + e] - c1 - h[4 * i + e] * c2);
    ((float4*)(dX + (size_t)b * T * D))[16 * i + c] = make_float4(t[0], t[1], t[2], t[3]);
    ((bf16x4*)(dXb + (size_t)b * T * D))[16 * i + c] = pack4(t);
  }
}

// Both kernels below sum over the batch: a chain of dependent loads whose LENGTH sets the time, so a workgroup is 64
// elements x 4 batch slices (one wave per slice), combined through LDS in a fixed order (deterministic).
// dgamma[d] = sum_b dfeat[b,d] xhat[b,d];  dbeta[d] = sum_b dfeat[b,d]
__global__ __launch_bounds__(1024) void cls_ln_affine_grad_kernel(const float* __restrict__ dfeat, const float* __restrict__ xhat,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int B) {
  // 64 elements x 16 batch slices: at batch 256 every thread issues its 16 row pairs at once (one memory round trip
  // instead of four: 26 -> 7 us); the 16 partials are combined in a fixed order
  __shared__ float s_g[16][64], s_b[16][64];
  const int el = threadIdx.x & 63, bs = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + el;
  float sg = 0.f, sb = 0.f;
  if (d < D) {
#pragma unroll 16
    for (int b = bs; b < B; b += 16) {
      const float g = dfeat[(size_t)b * D + d];
      sg = fmaf(g, xhat[(size_t)b * D + d], sg); sb += g;
    }
  }
  s_g[bs][el] = sg; s_b[bs][el] = sb;
  __syncthreads();
  if (bs == 0 && d < D) {
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { tg += s_g[q][el]; tb += s_b[q][el]; }
    dgamma[d] = tg;
    dbeta[d] = tb;
  }
}

// dpos[t,d] = sum_b dX[b,t,d];  dcls[d] = dpos[0,d]
template <typename TX>      // float: the fp32 dX; bf16 (round 4): the bf16 residual gradient that left block 0
__global__ __launch_bounds__(256) void pos_grad_kernel(const TX* __restrict__ dX, float* __restrict__ dpos,
                                                       float* __restrict__ dcls, int B, int T) {
  __shared__ float s_p[4][64];
  const int el = threadIdx.x & 63, bs = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < T * D) {
#pragma unroll 16
    for (int b = bs; b < B; b += 4) s += (float)dX[(size_t)b * T * D + e];
  }
  s_p[bs][el] = s;
  __syncthreads();
  if (bs == 0 && e < T * D) {
    const float t = (s_p[0][el] + s_p[1][el]) + (s_p[2][el] + s_p[3][el]);
    dpos[e] = t;
    if (e < D) dcls[e] = t;
  }
}

// ---- weight preparation ------------------------------------------------------------------------------------
// Wf[n][k] = bf16(W[n][k] * gamma[k]);  WfT[k][n] = same, transposed;  bias_f[n] = b[n] + sum_k W[n][k] beta[k]
// gamma/beta NULL: plain cast (+ transposed copy);  WT NULL: no transposed copy.
__global__ __launch_bounds__(256) void prep_weight_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16* __restrict__ Wf, bf16* __restrict__ WfT,
                                                          float* __restrict__ bias_f, int N, int K) {
  // one 16-lane group per output row n
  const int n = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = threadIdx.x & 15;
  if (n >= N) return;
  float dot = 0.f;
  for (int k = c; k < K; k += 16) {
    const float w = W[(size_t)n * K + k];
    const float wf = gamma ? w * gamma[k] : w;
    if (beta) dot = fmaf(w, beta[k], dot);
    Wf[(size_t)n * K + k] = (bf16)wf;
    if (WfT) WfT[(size_t)k * N + n] = (bf16)wf;
  }
  if (bias_f) {
    dot = wave_sum16(dot);
    if (c == 0) bias_f[n] = (bias ? bias[n] : 0.f) + dot;
  }
}

struct PrepBatch { RovitPrepDesc d[ROVIT_PREP_BATCH]; int first_group[ROVIT_PREP_BATCH + 1]; int n; };

// 16 weight rows per workgroup.  The transposed copy goes through an LDS tile so that it is written as 32-byte
// segments (16 consecutive n for one k) instead of 2-byte scatters at stride N (that version took 65 us per step).
__global__ __launch_bounds__(256) void prep_weight_batch_kernel(const PrepBatch pb) {
  constexpr int KMAX = 768, TS = KMAX + 8;
  __shared__ __attribute__((aligned(16))) bf16 tile[16 * TS];
  int i = 0;
  while (i + 1 < pb.n && (int)blockIdx.x >= pb.first_group[i + 1]) ++i;
  const RovitPrepDesc& d = pb.d[i];
  const int n0 = ((int)blockIdx.x - pb.first_group[i]) * 16;
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int n = n0 + r;
  bf16* Wf = (bf16*)d.Wf;
  bf16* WfT = (bf16*)d.WfT;
  const bool staged = WfT && d.K <= KMAX && d.K % 4 == 0;
  float dot = 0.f;
  if (n < d.N) {
    if (d.K % 4 == 0) {
      for (int k4 = c; k4 < d.K / 4; k4 += 16) {
        const float4 w = *(const float4*)(d.W + (size_t)n * d.K + 4 * k4);
        float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
        if (d.gamma) g = *(const float4*)(d.gamma + 4 * k4);
        if (d.beta) {
          const float4 be = *(const float4*)(d.beta + 4 * k4);
          dot = fmaf(w.x, be.x, dot); dot = fmaf(w.y, be.y, dot); dot = fmaf(w.z, be.z, dot); dot = fmaf(w.w, be.w, dot);
        }
        f32x4 wf = {w.x * g.x, w.y * g.y, w.z * g.z, w.w * g.w};
        const bf16x4 pk = pack4(wf);
        *(bf16x4*)(Wf + (size_t)n * d.K + 4 * k4) = pk;
        if (staged) *(bf16x4*)(tile + r * TS + 4 * k4) = pk;
        else if (WfT) { for (int e = 0; e < 4; ++e) WfT[(size_t)(4 * k4 + e) * d.N + n] = pk[e]; }
      }
    } else {
      for (int k = c; k < d.K; k += 16) {
        const float w = d.W[(size_t)n * d.K + k];
        const float wf = d.gamma ? w * d.gamma[k] : w;
        if (d.beta) dot = fmaf(w, d.beta[k], dot);
        Wf[(size_t)n * d.K + k] = (bf16)wf;
        if (WfT) WfT[(size_t)k * d.N + n] = (bf16)wf;
      }
    }
  }
  if (d.bias_f) {
    dot = wave_sum16(dot);
    if (c == 0 && n < d.N) d.bias_f[n] = (d.bias ? d.bias[n] : 0.f) + dot;
  }
  if (staged) {                                   // block-uniform
    __syncthreads();
    const int rows = min(16, d.N - n0);
    for (int k = threadIdx.x; k < d.K; k += 256) {
      bf16 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = tile[q * TS + k];
      bf16* dst = WfT + (size_t)k * d.N + n0;
      if (rows == 16 && (d.N % 8) == 0) {
        bf16x8 lo, hi;
#pragma unroll
        for (int q = 0; q < 8; ++q) { lo[q] = v[q]; hi[q] = v[8 + q]; }
        *(bf16x8*)dst = lo;
        *(bf16x8*)(dst + 8) = hi;
      } else {
        for (int q = 0; q < rows; ++q) dst[q] = v[q];
      }
    }
  }
}

}  // namespace

int rovit_prep_weight_batch(const RovitPrepDesc* descs, int n, rovit_stream_t stream) {
  for (int off = 0; off < n; off += ROVIT_PREP_BATCH) {
    PrepBatch pb{};
    pb.n = n - off < ROVIT_PREP_BATCH ? n - off : ROVIT_PREP_BATCH;
    int groups = 0;
    for (int i = 0; i < pb.n; ++i) {
      pb.d[i] = descs[off + i];
      pb.first_group[i] = groups;
      groups += (pb.d[i].N + 15) / 16;
    }
    pb.first_group[pb.n] = groups;
    hipLaunchKernelGGL(prep_weight_batch_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, pb);
    ROVIT_CHECK_LAUNCH("prep_weight_batch_kernel");
  }
  return ROVIT_OK;
}

extern "C" int rovit_layernorm_fwd(const float* x, void* xhat, float* rstd, int rows, int dim, float eps, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && xhat && rstd, ROVIT_ERR_NULL, "layernorm_fwd: null pointer");
  ROVIT_CHECK_ARG(dim == D && rows > 0, ROVIT_ERR_SHAPE, "layernorm_fwd: dim must be %d (got %d)", D, dim);
  ROVIT_CHECK_ARG(rovit_aligned16(x) && rovit_aligned16(xhat), ROVIT_ERR_ALIGN, "layernorm_fwd: alignment");
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, (bf16*)xhat, rstd, rows, eps,
                     (size_t)D, (size_t)D, (size_t)1);
  ROVIT_CHECK_LAUNCH("ln_fwd_kernel");
  return ROVIT_OK;
}

// same on every `row_step`-th row of the dense buffers (the CLS rows: row_step = tokens); results stay in place
int rovit_layernorm_fwd_rows(const float* x, void* xhat, float* rstd, int rows, int row_step, float eps, rovit_stream_t stream) {
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, (bf16*)xhat, rstd, rows, eps,
                     (size_t)D * row_step, (size_t)D * row_step, (size_t)row_step);
  ROVIT_CHECK_LAUNCH("ln_fwd_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_layernorm_bwd(const void* dxhat, const void* xhat, const float* rstd, float* dX, void* dXb, int rows,
                                   int dim, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dxhat && xhat && rstd && dX && dXb, ROVIT_ERR_NULL, "layernorm_bwd: null pointer");
  ROVIT_CHECK_ARG(dim == D && rows > 0, ROVIT_ERR_SHAPE, "layernorm_bwd: dim must be %d", D);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const bf16*)dxhat,
                     (const bf16*)xhat, rstd, dX, (bf16*)dXb, rows, (size_t)D, (size_t)1);
  ROVIT_CHECK_LAUNCH("ln_bwd_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_im2col(const float* x, void* col, int batch, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && col && batch > 0, ROVIT_ERR_NULL, "im2col: null pointer");
  ROVIT_CHECK_ARG(rovit_aligned16(x) && rovit_aligned16(col), ROVIT_ERR_ALIGN, "im2col: alignment");
  const size_t total = (size_t)batch * 196 * 96;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)col, batch);
  ROVIT_CHECK_LAUNCH("im2col_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_cls_rows(const float* cls, const float* pos, float* X, int batch, int tokens, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(cls && pos && X, ROVIT_ERR_NULL, "cls_rows: null pointer");
  hipLaunchKernelGGL(cls_row_kernel, dim3((batch * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos, X, batch, tokens);
  ROVIT_CHECK_LAUNCH("cls_row_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_cls_norm_fwd(const float* X, const float* gamma, const float* beta, float* feat, float* xhat,
                                  float* rstd, int batch, int tokens, float eps, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(X && gamma && beta && feat, ROVIT_ERR_NULL, "cls_norm_fwd: null pointer");
  hipLaunchKernelGGL(cls_ln_fwd_kernel, dim3((batch + 15) / 16), dim3(256), 0, (hipStream_t)stream, X, gamma, beta, feat, xhat,
                     rstd, batch, tokens, eps);
  ROVIT_CHECK_LAUNCH("cls_ln_fwd_kernel");
  return ROVIT_OK;
}

// writes the CLS rows of dX / dXb (zero_fill != 0: after zero-filling both for all tokens); also dgamma / dbeta of the final norm.
// rovit_vit_backward passes zero_fill = 0 (round 4): the last block's post-attention half reads CLS rows only, and the residual
// gradient that the other rows need (zero) is the memset of ONE bf16 buffer there instead of 57 MB of fills here.
extern "C" int rovit_cls_norm_bwd(const float* dfeat, const float* xhat, const float* rstd, const float* gamma, float* dX,
                                  void* dXb, float* dgamma, float* dbeta, int batch, int tokens, int zero_fill, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dfeat && xhat && rstd && gamma && dX && dXb, ROVIT_ERR_NULL, "cls_norm_bwd: null pointer");
  if (zero_fill) {
    const size_t n = (size_t)batch * tokens * D;
    hipError_t e1 = hipMemsetAsync(dX, 0, n * sizeof(float), (hipStream_t)stream);
    hipError_t e2 = hipMemsetAsync(dXb, 0, n * sizeof(bf16), (hipStream_t)stream);
    ROVIT_CHECK_ARG(e1 == hipSuccess && e2 == hipSuccess, ROVIT_ERR_LAUNCH, "cls_norm_bwd: memset failed");
  }
  hipLaunchKernelGGL(cls_ln_bwd_kernel, dim3((batch + 15) / 16), dim3(256), 0, (hipStream_t)stream, dfeat, xhat, rstd, gamma, dX,
                     (bf16*)dXb, batch, tokens);
  ROVIT_CHECK_LAUNCH("cls_ln_bwd_kernel");
  if (dgamma) {
    hipLaunchKernelGGL(cls_ln_affine_grad_kernel, dim3((D + 63) / 64), dim3(1024), 0, (hipStream_t)stream, dfeat, xhat, dgamma, dbeta, batch);
    ROVIT_CHECK_LAUNCH("cls_ln_affine_grad_kernel");
  }
  return ROVIT_OK;
}

// (internal, common.h) dgamma / dbeta of the final norm alone: sample sums nobody on the dgrad chain waits for -- rovit_vit_backward runs
// them on its weight-gradient stream
int rovit_cls_norm_affine_grad(const float* dfeat, const float* xhat, float* dgamma, float* dbeta, int batch, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dfeat && xhat && dgamma && dbeta, ROVIT_ERR_NULL, "cls_norm_affine_grad: null pointer");
  hipLaunchKernelGGL(cls_ln_affine_grad_kernel, dim3((D + 63) / 64), dim3(1024), 0, (hipStream_t)stream, dfeat, xhat, dgamma, dbeta, batch);
  ROVIT_CHECK_LAUNCH("cls_ln_affine_grad_kernel");
  return ROVIT_OK;
}

// exactly one of dX (fp32) / dXb (bf16, round 4) is the gradient w.r.t. the embedded tokens (B*tokens, 192)
extern "C" int rovit_pos_grad(const float* dX, const void* dXb, float* dpos, float* dcls, int batch, int tokens, rovit_stream_t stream) {
  ROVIT_CHECK_ARG((dX != nullptr) != (dXb != nullptr) && dpos && dcls, ROVIT_ERR_NULL, "pos_grad: pass exactly one of dX / dXb, and the outputs");
  const dim3 grid((tokens * D + 63) / 64), block(256);
  if (dX) hipLaunchKernelGGL(pos_grad_kernel<float>, grid, block, 0, (hipStream_t)stream, dX, dpos, dcls, batch, tokens);
  else hipLaunchKernelGGL(pos_grad_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)dXb, dpos, dcls, batch, tokens);
  ROVIT_CHECK_LAUNCH("pos_grad_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_prep_weight(const float* W, const float* bias, const float* gamma, const float* beta, void* Wf, void* WfT,
                                 float* bias_f, int N, int K, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(W && Wf, ROVIT_ERR_NULL, "prep_weight: null pointer");
  hipLaunchKernelGGL(prep_weight_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, W, bias, gamma, beta, (bf16*)Wf,
                     (bf16*)WfT, bias_f, N, K);
  ROVIT_CHECK_LAUNCH("prep_weight_kernel");
  return ROVIT_OK;
}
