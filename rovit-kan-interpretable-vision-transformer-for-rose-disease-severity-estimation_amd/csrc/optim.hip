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

// ---- the whole clip_grad_norm_ in ONE launch (round 4): squared norm over up to four buffers, then the coefficient ----
// Blocks walk the concatenation of the buffers in 64 KB pieces (4 x 16-byte loads in flight per thread, 1 024 threads); block partials go to
// scratch[8 + block] and the block whose ticket comes last adds them IN BLOCK ORDER (bit-reproducible), writes the norm and
// min(1, max_norm / (norm + 1e-6)) and re-arms the ticket.  Replaces fill + sq_norm x 2 + clip_coef (4 launches, ~37 us of which
// ~20 were one under-parallelised reduction over the backbone's 22 MB).
struct NormSegs { const float* g[4]; unsigned long long n4[4]; unsigned first[5]; int nseg; };     // first[]: in pieces
constexpr int NORM_PIECE4 = 4096;          // float4 per piece (64 KB: 1 024 threads x four 16-byte loads in flight)
constexpr int NORM_THREADS = 1024;
constexpr int NORM_MAX_BLOCKS = 256;       // one ticket atomic per block on ONE address: 1 343 blocks (one per piece) took 50 us, 256 take 8
__global__ __launch_bounds__(NORM_THREADS) void sq_norm_clip_kernel(const NormSegs sg, float max_norm, float* __restrict__ coef,
                                                           float* __restrict__ norm_out, float* __restrict__ scratch) {
  __shared__ float s_part[NORM_THREADS / 64];
  __shared__ bool s_last;
  const unsigned pieces = sg.first[sg.nseg];
  const unsigned p0 = (unsigned)((unsigned long long)blockIdx.x * pieces / gridDim.x), p1 = (unsigned)((unsigned long long)(blockIdx.x + 1) * pieces / gridDim.x);
  float acc = 0.f;
  for (unsigned pc = p0; pc < p1; ++pc) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (q < sg.nseg && pc >= sg.first[q]) k = q;
    const float4* g = (const float4*)(k == 0 ? sg.g[0] : (k == 1 ? sg.g[1] : (k == 2 ? sg.g[2] : sg.g[3])));
    const unsigned long long n4 = k == 0 ? sg.n4[0] : (k == 1 ? sg.n4[1] : (k == 2 ? sg.n4[2] : sg.n4[3]));
    const unsigned f0 = k == 0 ? sg.first[0] : (k == 1 ? sg.first[1] : (k == 2 ? sg.first[2] : sg.first[3]));
    const unsigned long long base = (unsigned long long)(pc - f0) * NORM_PIECE4;
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long i = base + u * NORM_THREADS + threadIdx.x;
      v[u] = i < n4 ? g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
  }
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NORM_THREADS / 64; ++w) t += s_part[w];          // waves in order
    scratch[8 + blockIdx.x] = t;
    __threadfence();
    const unsigned ticket = atomicAdd((unsigned*)scratch, 1u);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {                                   // block-uniform
    __threadfence();
    float t = 0.f;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += NORM_THREADS) t += __builtin_nontemporal_load(scratch + 8 + b);
    t = wave_sum64(t);                            // fixed-shape tree over (thread, wave): independent of which block came last
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tt = 0.f;
#pragma unroll
      for (int w = 0; w < NORM_THREADS / 64; ++w) tt += s_part[w];
      const float n = sqrtf(tt);
      if (norm_out) *norm_out = n;
      *coef = fminf(1.f, max_norm / (n + 1e-6f));
      *(unsigned*)scratch = 0u;                   // ready for the next call on this stream
    }
  }
}

// ---- AdamW over up to four flat segments (parameter groups / step counts of their own) in one launch ----
struct AdamSegs { float* p[4]; const float* g[4]; float* m[4]; float* v[4]; unsigned long long n[4]; float lr[4], inv_bc1[4], inv_sqrt_bc2[4];
                  unsigned first[5]; int nseg; };
constexpr int ADAM_PIECE4 = 512;           // float4 per block
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamSegs sg, const float* __restrict__ grad_scale, float beta1, float beta2,
                                                          float eps, float wd) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q < sg.nseg && blockIdx.x >= sg.first[q]) k = q;
#define SEL(a) (k == 0 ? sg.a[0] : (k == 1 ? sg.a[1] : (k == 2 ? sg.a[2] : sg.a[3])))
  float* p = SEL(p); const float* g = SEL(g); float* m = SEL(m); float* v = SEL(v);
  const unsigned long long n = SEL(n);
  const float lr = SEL(lr), inv_bc1 = SEL(inv_bc1), inv_sqrt_bc2 = SEL(inv_sqrt_bc2);
  const unsigned f0 = SEL(first);
#undef SEL
  const float gs = grad_scale ? *grad_scale : 1.f;
  const unsigned long long n4 = n / 4;
  const float decay = 1.f - lr * wd;
  const float step = lr * inv_bc1;
  const unsigned long long base = (unsigned long long)(blockIdx.x - f0) * ADAM_PIECE4;
#pragma unroll
  for (int u = 0; u < ADAM_PIECE4 / 256; ++u) {
    const unsigned long long i = base + u * 256 + threadIdx.x;
    if (i < n4) {
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
  }
  if (blockIdx.x == f0) {
    for (unsigned long long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gs;
      m[i] = beta1 * m[i] + (1.f - beta1) * gr;
      v[i] = beta2 * v[i] + (1.f - beta2) * gr * gr;
      p[i] = p[i] * decay - step * (m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps));
    }
  }
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

// clip_grad_norm_ over up to four gradient buffers in one launch: *norm_out = sqrt(sum of squares) (optional),
// *coef = min(1, max_norm / (norm + 1e-6)).  bufs / counts: HOST arrays; every buffer 16-byte aligned with a count that is a multiple
// of 4 (the flat buffers are padded so).  scratch: at least 264 floats, zeroed once by the caller.
extern "C" int rovit_sq_norm_clip(const float* const* bufs, const size_t* counts, int n_bufs, float max_norm, float* coef, float* norm_out,
                                  float* scratch, size_t scratch_floats, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(bufs && counts && coef && scratch, ROVIT_ERR_NULL, "sq_norm_clip: null pointer");
  ROVIT_CHECK_ARG(n_bufs >= 1 && n_bufs <= 4, ROVIT_ERR_SHAPE, "sq_norm_clip: 1..4 buffers, got %d", n_bufs);
  ROVIT_CHECK_ARG(max_norm > 0.f, ROVIT_ERR_SHAPE, "sq_norm_clip: max_norm must be positive");
  NormSegs sg{};
  unsigned blocks = 0;
  for (int i = 0; i < n_bufs; ++i) {
    ROVIT_CHECK_ARG(bufs[i] && rovit_aligned16(bufs[i]) && counts[i] % 4 == 0, ROVIT_ERR_ALIGN,
                    "sq_norm_clip: buffer %d must be 16-byte aligned with a multiple of 4 floats", i);
    sg.g[i] = bufs[i]; sg.n4[i] = counts[i] / 4; sg.first[i] = blocks;
    blocks += (unsigned)((counts[i] / 4 + NORM_PIECE4 - 1) / NORM_PIECE4);
  }
  sg.first[n_bufs] = blocks; sg.nseg = n_bufs;
  ROVIT_CHECK_ARG(blocks >= 1, ROVIT_ERR_SHAPE, "sq_norm_clip: empty buffers");
  const unsigned grid = blocks < (unsigned)NORM_MAX_BLOCKS ? blocks : (unsigned)NORM_MAX_BLOCKS;       // pieces are dealt to the blocks in order
  ROVIT_CHECK_ARG((size_t)grid + 8 <= scratch_floats, ROVIT_ERR_SHAPE, "sq_norm_clip: scratch holds %zu floats, %u needed", scratch_floats, grid + 8);
  hipLaunchKernelGGL(sq_norm_clip_kernel, dim3(grid), dim3(NORM_THREADS), 0, (hipStream_t)stream, sg, max_norm, coef, norm_out, scratch);
  ROVIT_CHECK_LAUNCH("sq_norm_clip_kernel");
  return ROVIT_OK;
}

// rovit_adamw_flat over up to four segments in one launch; lr / t per segment (parameter groups, curriculum-gated step counts).
extern "C" int rovit_adamw_flat_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const size_t* n,
                                      const float* lr, const int* t, int n_segs, const float* grad_scale, float beta1, float beta2, float eps,
                                      float weight_decay, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(p && g && m && v && n && lr && t, ROVIT_ERR_NULL, "adamw_flat_multi: null pointer");
  ROVIT_CHECK_ARG(n_segs >= 1 && n_segs <= 4, ROVIT_ERR_SHAPE, "adamw_flat_multi: 1..4 segments, got %d", n_segs);
  AdamSegs sg{};
  unsigned blocks = 0;
  for (int i = 0; i < n_segs; ++i) {
    ROVIT_CHECK_ARG(p[i] && g[i] && m[i] && v[i], ROVIT_ERR_NULL, "adamw_flat_multi: null pointer in segment %d", i);
    ROVIT_CHECK_ARG(t[i] >= 1, ROVIT_ERR_SHAPE, "adamw_flat_multi: step count must be >= 1");
    ROVIT_CHECK_ARG(rovit_aligned16(p[i]) && rovit_aligned16(g[i]) && rovit_aligned16(m[i]) && rovit_aligned16(v[i]), ROVIT_ERR_ALIGN,
                    "adamw_flat_multi: buffers of segment %d must be 16-byte aligned", i);
    const double bc1 = 1.0 - pow((double)beta1, t[i]), bc2 = 1.0 - pow((double)beta2, t[i]);
    sg.p[i] = p[i]; sg.g[i] = g[i]; sg.m[i] = m[i]; sg.v[i] = v[i]; sg.n[i] = n[i];
    sg.lr[i] = lr[i]; sg.inv_bc1[i] = (float)(1.0 / bc1); sg.inv_sqrt_bc2[i] = (float)(1.0 / sqrt(bc2));
    sg.first[i] = blocks;
    const unsigned b = (unsigned)((n[i] / 4 + ADAM_PIECE4 - 1) / ADAM_PIECE4);
    blocks += b < 1 ? 1 : b;
  }
  sg.first[n_segs] = blocks; sg.nseg = n_segs;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sg, grad_scale, beta1, beta2, eps, weight_decay);
  ROVIT_CHECK_LAUNCH("adamw_multi_kernel");
  return ROVIT_OK;
}
