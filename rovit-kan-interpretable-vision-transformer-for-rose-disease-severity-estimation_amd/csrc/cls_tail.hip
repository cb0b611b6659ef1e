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
