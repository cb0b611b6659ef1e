// The last block's post-attention half and the final norm on the class-token rows, ONE launch each way (round 4).
//
// Reference arithmetic being restated: timm Block (x = x + proj(attn); x = x + mlp(norm2(x))) followed by VisionTransformer.norm and
// forward_head's x[:, 0], reached through /root/reference/models/backbone.py:23-25.  Only token 0 of the last block's output is consumed,
// and everything behind the attention is row-wise, so since round 1 this half runs on the B class-token rows alone -- as a chain of six
// small launches forward (proj + residual, norm2, fc1 + GELU, fc2 + residual, final norm: 57 us for 256 rows, each a dependent launch that
// leaves the chip empty) and five backward.  Here ONE workgroup owns ONE row for the whole chain, like the head phase (head_phase.hip):
// the 0.66 MB of bf16 weight images are streamed from L2 once per workgroup (4 lanes / 16 lanes per output row, 16-byte loads), the row
// lives in LDS between the stages.  The arithmetic mirrors the launches it replaces: bf16 operands with fp32 accumulation, xhat2 and the
// pre-activation rounded to bf16 where the GEMM path stages them, GELU by gelu_and_grad on the bf16-rounded input, act / gelu' kept in bf16
// for the class-token weight gradients (which still run as one merged launch on the weight-gradient stream).
#include "common.h"

namespace {

constexpr int D = 192, MLP = 768;

struct ClsTailArgs {
  const bf16* o; long o_ld;              // attention output: row b at o + b * o_ld
  float* X; long x_ld;                   // residual stream (fp32), updated in place
  const bf16* wproj; const float* bproj; // (192,192) bf16 image, fp32 bias
  const bf16* wfc1; const float* bfc1;   // (768,192) with norm2's affine folded in, folded bias
  const bf16* wfc2; const float* bfc2;   // (192,768)
  const float* gamma; const float* beta; // final norm
  bf16* xhat2; long xh_ld; float* rstd2; long rs_ld;     // kept for the backward (NULL: inference)
  bf16* act; bf16* dact; long act_ld;
  float* feat; float* xhat_cls; float* rstd_cls;          // (nb,192), (nb,192) or NULL, (nb) or NULL
  float eps;
};

__device__ __forceinline__ float dot8(const bf16x8 w, const float* x) {
  float t = (float)w[0] * x[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) t = fmaf((float)w[e], x[e], t);
  return t;
}

__global__ __launch_bounds__(1024) void cls_tail_fwd_kernel(const ClsTailArgs a) {
  __shared__ __attribute__((aligned(16))) float s_o[D], s_x[D], s_h[D], s_act[MLP];
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63;
  if (tid < D) {
    s_o[tid] = (float)a.o[(size_t)b * a.o_ld + tid];
    s_x[tid] = a.X[(size_t)b * a.x_ld + tid];
  }
  __syncthreads();
  // proj + bias + residual: 192 rows x 192, four lanes per row
  if (tid < 4 * D) {
    const int row = tid >> 2, part = tid & 3;
    const bf16x8* w = (const bf16x8*)(a.wproj + (size_t)row * D);
    float acc = 0.f;
#pragma unroll
    for (int q = part; q < D / 8; q += 4) acc += dot8(w[q], s_o + 8 * q);
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0) s_x[row] += acc + a.bproj[row];
  }
  __syncthreads();
  // norm2 (one wave; its affine lives in the folded fc1 weight): xhat2 rounded to bf16, as the GEMM path stages it
  if (tid < 64) {
    const float v0 = s_x[lane], v1 = s_x[lane + 64], v2 = s_x[lane + 128];
    const float mean = wave_sum64(v0 + v1 + v2) * (1.f / D);
    const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean;
    const float r = rsqrtf(wave_sum64(d0 * d0 + d1 * d1 + d2 * d2) * (1.f / D) + a.eps);
    const bf16 h0 = (bf16)(d0 * r), h1 = (bf16)(d1 * r), h2 = (bf16)(d2 * r);
    s_h[lane] = (float)h0; s_h[lane + 64] = (float)h1; s_h[lane + 128] = (float)h2;
    if (a.xhat2) {
      bf16* xr = a.xhat2 + (size_t)b * a.xh_ld;
      xr[lane] = h0; xr[lane + 64] = h1; xr[lane + 128] = h2;
      if (lane == 0) a.rstd2[(size_t)b * a.rs_ld] = r;
    }
  }
  __syncthreads();
  // fc1 + bias + GELU: 768 rows x 192, four lanes per row, three passes
  for (int item = tid; item < 4 * MLP; item += 1024) {
    const int row = item >> 2, part = item & 3;
    const bf16x8* w = (const bf16x8*)(a.wfc1 + (size_t)row * D);
    float acc = 0.f;
#pragma unroll
    for (int q = part; q < D / 8; q += 4) acc += dot8(w[q], s_h + 8 * q);
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0) {
      float ga, gd;
      gelu_and_grad((float)(bf16)(acc + a.bfc1[row]), ga, gd);
      const bf16 ab = (bf16)ga;
      s_act[row] = (float)ab;
      if (a.act) a.act[(size_t)b * a.act_ld + row] = ab;
      if (a.dact) a.dact[(size_t)b * a.act_ld + row] = (bf16)gd;
    }
  }
  __syncthreads();
  // fc2 + bias + residual: 192 rows x 768, sixteen lanes per row, three passes
  for (int item = tid; item < 16 * D; item += 1024) {
    const int row = item >> 4, part = item & 15;
    const bf16x8* w = (const bf16x8*)(a.wfc2 + (size_t)row * MLP);
    float acc = 0.f;
#pragma unroll
    for (int q = part; q < MLP / 8; q += 16) acc += dot8(w[q], s_act + 8 * q);
    acc = wave_sum16(acc);
    if (part == 0) s_x[row] += acc + a.bfc2[row];
  }
  __syncthreads();
  // the final norm of the class token: features (and xhat / rstd for its backward)
  if (tid < 64) {
    const float v0 = s_x[lane], v1 = s_x[lane + 64], v2 = s_x[lane + 128];
    const float mean = wave_sum64(v0 + v1 + v2) * (1.f / D);
    const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean;
    const float r = rsqrtf(wave_sum64(d0 * d0 + d1 * d1 + d2 * d2) * (1.f / D) + a.eps);
    float* xr = a.X + (size_t)b * a.x_ld;
    xr[lane] = v0; xr[lane + 64] = v1; xr[lane + 128] = v2;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = lane + 64 * u;
      const float h = (u == 0 ? d0 : (u == 1 ? d1 : d2)) * r;
      a.feat[(size_t)b * D + c] = h * a.gamma[c] + a.beta[c];
      if (a.xhat_cls) a.xhat_cls[(size_t)b * D + c] = h;
    }
    if (lane == 0 && a.rstd_cls) a.rstd_cls[b] = r;
  }
}

// ---- backward of the same chain, one workgroup per row: final-norm backward -> fc2 dgrad x gelu' -> fc1 dgrad -> norm2 backward -> proj dgrad.
// Mirrors the launches it replaces (rovit_cls_norm_bwd, rovit_gemm_nt EPI_MUL / EPI_BF16, rovit_layernorm_bwd_rows): the residual gradient
// is rounded to bf16 where a GEMM reads it as its operand, dgrad results are rounded to bf16 where the GEMM path stages them, the norm
// backward sums in fp32.  The dgrad products contract over the ROWS of the row-major forward weight images (no transposed image needed):
// thread = (eight consecutive output columns, row slice), partial sums meet in LDS in slice order.
struct ClsTailBwdArgs {
  const float* dfeat; const float* xhat_cls; const float* rstd_cls; const float* gamma;      // final norm
  const bf16* wfc2; const bf16* wfc1; const bf16* wproj;                                      // (192,768), (768,192), (192,192)
  const bf16* dact; long act_ld;         // gelu' rows
  const bf16* xhat2; long xh_ld; const float* rstd2; long rs_ld;
  bf16* xin; bf16* dpre; bf16* xmid; bf16* dO; long row_ld192; long row_ld768;               // outputs, row b at b * row_ld (elements)
};

__global__ __launch_bounds__(1024) void cls_tail_bwd_kernel(const ClsTailBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_part[32 * D];          // [8][768] or [32][192] partial sums
  __shared__ __attribute__((aligned(16))) float s_dx[D], s_v[D], s_dp[MLP];
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63;
  // final-norm backward (one wave): dx0 = rstd (g - mean(g) - xhat mean(g xhat)), g = dfeat * gamma
  if (tid < 64) {
    float g[3], h[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = lane + 64 * u;
      g[u] = a.dfeat[(size_t)b * D + c] * a.gamma[c];
      h[u] = a.xhat_cls[(size_t)b * D + c];
      s1 += g[u]; s2 += g[u] * h[u];
    }
    const float c1 = wave_sum64(s1) * (1.f / D), c2 = wave_sum64(s2) * (1.f / D), r = a.rstd_cls[b];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = lane + 64 * u;
      const float dx = r * (g[u] - c1 - h[u] * c2);
      const bf16 xb = (bf16)dx;
      s_dx[c] = dx;                          // the fp32 residual gradient (norm2's backward adds onto it)
      s_v[c] = (float)xb;                    // its bf16 rounding: the fc2 dgrad's operand and the fc2 weight gradient's dY
      a.xin[(size_t)b * a.row_ld192 + c] = xb;
    }
  }
  __syncthreads();
  // fc2 dgrad: t[k] = sum_n xin[n] W2[n][k]; thread = (8 consecutive k, one of 8 slices of 24 rows n)
  if (tid < 768) {
    const int kc = tid % 96, sl = tid / 96;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
    for (int n = 24 * sl; n < 24 * sl + 24; ++n) {
      const bf16x8 w = *(const bf16x8*)(a.wfc2 + (size_t)n * MLP + 8 * kc);
      const float x = s_v[n];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(x, (float)w[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_part[sl * MLP + 8 * kc + e] = acc[e];
  }
  __syncthreads();
  if (tid < MLP) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) t += s_part[sl * MLP + tid];
    const bf16 dp = (bf16)((float)(bf16)t * (float)a.dact[(size_t)b * a.act_ld + tid]);      // x gelu', staged in bf16 like EPI_MUL
    s_dp[tid] = (float)dp;
    a.dpre[(size_t)b * a.row_ld768 + tid] = dp;
  }
  __syncthreads();
  // fc1 dgrad: dxhat2[i] = sum_k dpre[k] W1f[k][i]; thread = (8 consecutive i, one of 32 slices of 24 rows k)
  if (tid < 768) {
    const int ic = tid % 24, sl = tid / 24;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
    for (int k = 24 * sl; k < 24 * sl + 24; ++k) {
      const bf16x8 w = *(const bf16x8*)(a.wfc1 + (size_t)k * D + 8 * ic);
      const float x = s_dp[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(x, (float)w[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_part[sl * D + 8 * ic + e] = acc[e];
  }
  __syncthreads();
  if (tid < D) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) t += s_part[sl * D + tid];
    s_v[tid] = (float)(bf16)t;               // dxhat2, staged in bf16 like EPI_BF16
  }
  __syncthreads();
  // norm2 backward (one wave): dx1 = dx0 + rstd2 (v - mean(v) - xhat2 mean(v xhat2)); its bf16 rounding = the mid-block gradient
  if (tid < 64) {
    float v[3], h[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = lane + 64 * u;
      v[u] = s_v[c];
      h[u] = (float)a.xhat2[(size_t)b * a.xh_ld + c];
      s1 += v[u]; s2 += v[u] * h[u];
    }
    const float c1 = wave_sum64(s1) * (1.f / D), c2 = wave_sum64(s2) * (1.f / D), r = a.rstd2[(size_t)b * a.rs_ld];
    float dx[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) dx[u] = s_dx[lane + 64 * u] + r * (v[u] - c1 - h[u] * c2);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = lane + 64 * u;
      const bf16 xb = (bf16)dx[u];
      s_dx[c] = (float)xb;
      a.xmid[(size_t)b * a.row_ld192 + c] = xb;
    }
  }
  __syncthreads();
  // proj dgrad: dO[i] = sum_n xmid[n] Wp[n][i]; thread = (8 consecutive i, one of 32 slices of 6 rows n)
  if (tid < 768) {
    const int ic = tid % 24, sl = tid / 24;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 6 * sl; n < 6 * sl + 6; ++n) {
      const bf16x8 w = *(const bf16x8*)(a.wproj + (size_t)n * D + 8 * ic);
      const float x = s_dx[n];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(x, (float)w[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_part[sl * D + 8 * ic + e] = acc[e];
  }
  __syncthreads();
  if (tid < D) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) t += s_part[sl * D + tid];
    a.dO[(size_t)b * a.row_ld192 + tid] = (bf16)t;
  }
}

}  // namespace

// (internal, common.h) rows: the class-token rows of `rows` images, row b of every dense buffer at b * tokens rows
int rovit_cls_tail_fwd(const void* o, float* X, const void* wproj, const float* bproj, const void* wfc1, const float* bfc1, const void* wfc2,
                       const float* bfc2, const float* gamma, const float* beta, void* xhat2, float* rstd2, void* act, void* dact, float* feat,
                       float* xhat_cls, float* rstd_cls, int rows, int tokens, float eps, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(o && X && wproj && bproj && wfc1 && bfc1 && wfc2 && bfc2 && gamma && beta && feat, ROVIT_ERR_NULL, "cls_tail_fwd: null pointer");
  ROVIT_CHECK_ARG((!xhat2) == (!rstd2) && (act || !dact), ROVIT_ERR_NULL, "cls_tail_fwd: xhat2 / rstd2 come together, dact needs act");
  ROVIT_CHECK_ARG(rows > 0 && tokens > 0, ROVIT_ERR_SHAPE, "cls_tail_fwd: bad shape");
  ROVIT_CHECK_ARG(rovit_aligned16(wproj) && rovit_aligned16(wfc1) && rovit_aligned16(wfc2), ROVIT_ERR_ALIGN, "cls_tail_fwd: weight images must be 16-byte aligned");
  ClsTailArgs a{};
  a.o = (const bf16*)o; a.o_ld = (long)tokens * D; a.X = X; a.x_ld = (long)tokens * D;
  a.wproj = (const bf16*)wproj; a.bproj = bproj; a.wfc1 = (const bf16*)wfc1; a.bfc1 = bfc1; a.wfc2 = (const bf16*)wfc2; a.bfc2 = bfc2;
  a.gamma = gamma; a.beta = beta;
  a.xhat2 = (bf16*)xhat2; a.xh_ld = (long)tokens * D; a.rstd2 = rstd2; a.rs_ld = tokens;
  a.act = (bf16*)act; a.dact = (bf16*)dact; a.act_ld = (long)tokens * MLP;
  a.feat = feat; a.xhat_cls = xhat_cls; a.rstd_cls = rstd_cls; a.eps = eps;
  hipLaunchKernelGGL(cls_tail_fwd_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("cls_tail_fwd_kernel");
  return ROVIT_OK;
}

// (internal, common.h) the backward chain of the same rows: writes the class-token rows of xin (bf16 gradient entering the block), dpre,
// xmid (mid-block gradient) and dO
int rovit_cls_tail_bwd(const float* dfeat, const float* xhat_cls, const float* rstd_cls, const float* gamma, const void* wfc2, const void* wfc1,
                       const void* wproj, const void* dact, const void* xhat2, const float* rstd2, void* xin, void* dpre, void* xmid, void* dO,
                       int rows, int tokens, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dfeat && xhat_cls && rstd_cls && gamma && wfc2 && wfc1 && wproj && dact && xhat2 && rstd2 && xin && dpre && xmid && dO,
                  ROVIT_ERR_NULL, "cls_tail_bwd: null pointer");
  ROVIT_CHECK_ARG(rows > 0 && tokens > 0, ROVIT_ERR_SHAPE, "cls_tail_bwd: bad shape");
  ROVIT_CHECK_ARG(rovit_aligned16(wproj) && rovit_aligned16(wfc1) && rovit_aligned16(wfc2), ROVIT_ERR_ALIGN, "cls_tail_bwd: weight images must be 16-byte aligned");
  ClsTailBwdArgs a{};
  a.dfeat = dfeat; a.xhat_cls = xhat_cls; a.rstd_cls = rstd_cls; a.gamma = gamma;
  a.wfc2 = (const bf16*)wfc2; a.wfc1 = (const bf16*)wfc1; a.wproj = (const bf16*)wproj;
  a.dact = (const bf16*)dact; a.act_ld = (long)tokens * MLP; a.xhat2 = (const bf16*)xhat2; a.xh_ld = (long)tokens * D; a.rstd2 = rstd2; a.rs_ld = tokens;
  a.xin = (bf16*)xin; a.dpre = (bf16*)dpre; a.xmid = (bf16*)xmid; a.dO = (bf16*)dO; a.row_ld192 = (long)tokens * D; a.row_ld768 = (long)tokens * MLP;
  hipLaunchKernelGGL(cls_tail_bwd_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("cls_tail_bwd_kernel");
  return ROVIT_OK;
}
