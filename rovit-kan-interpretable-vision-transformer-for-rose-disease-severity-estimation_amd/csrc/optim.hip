// Optimizer step on the flat parameter / gradient buffers of the backbone.
//
// Reference being restated: torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0) followed by AdamW
// (/root/reference/training/trainer.py:123-128,137-141; training/optimizer.py:7-32: AdamW, weight_decay 1e-4,
// backbone group at lr/10).  All 5.5 M backbone parameters live in ONE contiguous fp32 buffer, so the whole
// update is two launches: a squared-norm reduction and a fused (clip-scale + decoupled weight decay + Adam) pass
// reading p, g, m, v once and writing p, m, v once.
#include "common.h"

namespace {

// scratch (optional): [0] ticket counter (as unsigned, left at 0), [8 .. 8+gridDim) per-block partial sums.  With a
// scratch buffer the block partials are added in block order by the block that finishes last, so the result does not
// depend on scheduling (bit-reproducible training steps); without it they are added with float atomics.
__global__ __launch_bounds__(256) void sq_norm_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out,
                                                      float* __restrict__ scratch) {
  __shared__ float s_part[4];
  __shared__ bool s_last;
  float acc = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = ((const float4*)g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (!scratch) {
    if (threadIdx.x == 0) atomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
    return;
  }
  if (threadIdx.x == 0) {
    scratch[8 + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    __threadfence();
    const unsigned ticket = atomicAdd((unsigned*)scratch, 1u);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {                                   // block-uniform
    __threadfence();
    float t = 0.f;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __builtin_nontemporal_load(scratch + 8 + b);
    // fixed-shape tree over (thread, wave): independent of which block came last
    t = wave_sum64(t);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      *out += (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
      *(unsigned*)scratch = 0u;                   // ready for the next call on this stream
    }
  }
}

// p, g, m, v: n floats.  grad_scale: device scalar multiplied into g (the clip coefficient), or NULL.
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, size_t n,
                                                         const float* __restrict__ grad_scale, float lr, float beta1,
                                                         float beta2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
  const float gs = grad_scale ? *grad_scale : 1.f;
  const size_t n4 = n / 4;
  const float decay = 1.f - lr * wd;
  const float step = lr * inv_bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    const float4 gg = ((const float4*)g)[i];
    float* P = (float*)&pp; float* M = (float*)&mm; float* V = (float*)&vv; const float* G = (const float*)&gg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = G[e] * gs;
      M[e] = beta1 * M[e] + (1.f - beta1) * gr;
      V[e] = beta2 * V[e] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(V[e]) * inv_sqrt_bc2 + eps;
      P[e] = P[e] * decay - step * (M[e] / denom);
    }
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gs;
      m[i] = beta1 * m[i] + (1.f - beta1) * gr;
      v[i] = beta2 * v[i] + (1.f - beta2) * gr * gr;
      p[i] = p[i] * decay - step * (m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps));
    }
  }
}

// clip_grad_norm_ coefficient from the accumulated squared norm: norm = sqrt(sq), coef = min(1, max_norm / (norm + 1e-6))
__global__ void clip_coef_kernel(const float* __restrict__ sq, float max_norm, float* __restrict__ coef, float* __restrict__ norm_out) {
  const float n = sqrtf(*sq);
  if (norm_out) *norm_out = n;
  *coef = fminf(1.f, max_norm / (n + 1e-6f));
}

}  // namespace

extern "C" int rovit_clip_coef(const float* sq, float max_norm, float* coef, float* norm_out, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(sq && coef, ROVIT_ERR_NULL, "clip_coef: null pointer");
  ROVIT_CHECK_ARG(max_norm > 0.f, ROVIT_ERR_SHAPE, "clip_coef: max_norm must be positive");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sq, max_norm, coef, norm_out);
  ROVIT_CHECK_LAUNCH("clip_coef_kernel");
  return ROVIT_OK;
}

// out_sq (device scalar) += sum g^2 ; the caller zeroes it (so several buffers can accumulate into one norm)
extern "C" int rovit_sq_norm_accum(const float* g, size_t n, float* out_sq, float* scratch, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(g && out_sq, ROVIT_ERR_NULL, "sq_norm_accum: null pointer");
  ROVIT_CHECK_ARG(rovit_aligned16(g), ROVIT_ERR_ALIGN, "sq_norm_accum: gradient buffer must be 16-byte aligned");
  if (n == 0) return ROVIT_OK;
  size_t blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);      // one atomic per block onto a single address
  hipLaunchKernelGGL(sq_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, out_sq, scratch);
  ROVIT_CHECK_LAUNCH("sq_norm_kernel");
  return ROVIT_OK;
}

// torch.optim.AdamW semantics (decoupled weight decay, bias correction by step count `t` >= 1)
extern "C" int rovit_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, const float* grad_scale, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int t, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(p && g && m && v, ROVIT_ERR_NULL, "adamw_flat: null pointer");
  ROVIT_CHECK_ARG(t >= 1, ROVIT_ERR_SHAPE, "adamw_flat: step count must be >= 1");
  ROVIT_CHECK_ARG(rovit_aligned16(p) && rovit_aligned16(g) && rovit_aligned16(m) && rovit_aligned16(v), ROVIT_ERR_ALIGN,
                  "adamw_flat: buffers must be 16-byte aligned");
  if (n == 0) return ROVIT_OK;
  const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
  size_t blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, grad_scale, lr,
                     beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
  ROVIT_CHECK_LAUNCH("adamw_flat_kernel");
  return ROVIT_OK;
}
