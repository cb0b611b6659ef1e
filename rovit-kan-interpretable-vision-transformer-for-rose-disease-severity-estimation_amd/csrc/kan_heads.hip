// KAN B-spline head and the three MLP heads: fp32 wavefront-level kernels (no MFMA).
//
// Reference semantics:
//   KANLayer.forward            /root/reference/models/kan.py:70-95   (tanh -> truncated cubic basis ->
//                                                                       spline contraction + Linear(x))
//   BSplineBasis.compute_basis  models/kan.py:8-44                     (closed form: SURVEY.md 8(a) addendum)
//   KANSeverityModule.forward   models/kan.py:138-149                  (ReLU between layers, 3*sigmoid at the end)
//   Classification/Ordinal/UncertaintyHead.forward  models/heads.py:17-22, 38-43, 91-102
//
// The (B, in, num_basis) basis tensor the reference materialises (kan.py:20) never exists here: per
// (sample, input feature) the kernel keeps the knot interval index and the 4 non-zero cubic values in LDS.
#include "common.h"
#include "kan_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// forward: one workgroup = TB samples.  Phase 1: per (sample, feature) tanh + grid lookup -> LDS.
// Phase 2: thread = (sample, output): 4 FMAs against the W[i, o, j-3..j] slab + the linear term.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void kan_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ knots, const float* __restrict__ lw,
                                                      const float* __restrict__ lb, float* __restrict__ out, int B,
                                                      int in_f, int out_f, int nk, int TB, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_knots = smem;
  float* s_x = s_knots + KAN_MAX_KNOTS;
  float* s_v = s_x + TB * in_f;
  int* s_j = (int*)(s_v + 4 * TB * in_f);
  const int tid = threadIdx.x, T = blockDim.x;
  const int nb = nk - 4;
  const int b0 = blockIdx.x * TB;
  if (tid < nk) s_knots[tid] = knots[tid];
  __syncthreads();
  const float inv_h0 = 1.f / (s_knots[1] - s_knots[0]);
  for (int e = tid; e < TB * in_f; e += T) {
    const int bl = e / in_f, i = e - bl * in_f, b = b0 + bl;
    float xv = 0.f;
    Basis4 bs; bs.j = -1; bs.v[0] = bs.v[1] = bs.v[2] = bs.v[3] = 0.f;
    if (b < B) {
      xv = x[(size_t)b * in_f + i];
      bs = kan_basis<false>(tanhf(xv), s_knots, nk, inv_h0, nullptr);
    }
    s_x[e] = xv;
    s_j[e] = bs.j;
    *(float4*)(s_v + 4 * e) = make_float4(bs.v[0], bs.v[1], bs.v[2], bs.v[3]);
  }
  __syncthreads();
  // Contraction.  The (sample, output) items of this workgroup are few (<= 64 for the layer shapes of this
  // model), so each item is shared by `nsplit` threads that each walk a slice of the input features (the loop is a
  // chain of dependent L2 gathers: its length sets the time.  1024 threads give 16 slices of 12 features for the 192 -> 64
  // layer at one sample per workgroup: 28 -> 14 us at batch 256).
  float* s_part = (float*)(s_j + TB * in_f);            // [T] partial sums
  const int items = TB * out_f;
  const int nsplit = items >= T ? 1 : T / items;
  for (int e0 = 0; e0 < items; e0 += T) {
    const int e = e0 + (nsplit == 1 ? tid : tid % items);
    const int part = nsplit == 1 ? 0 : tid / items;
    float acc = 0.f;
    const bool live = e < items && part < nsplit;
    const int bl = live ? e / out_f : 0, o = live ? e - bl * out_f : 0, b = b0 + bl;
    if (live && b < B) {
      const int chunk = (in_f + nsplit - 1) / nsplit;
      const int i_lo = part * chunk, i_hi = min(in_f, i_lo + chunk);
      const float* lwo = lw + (size_t)o * in_f;
      // branch-free gathers (clamped index, masked value) so that several features' loads are in flight at once:
      // the loop is a chain of L2 round trips, not arithmetic
#pragma unroll 8
      for (int i = i_lo; i < i_hi; ++i) {
        const int q = bl * in_f + i;
        const int j = s_j[q];
        const float4 v = *(const float4*)(s_v + 4 * q);
        const float* w = W + ((size_t)i * out_f + o) * nb;
        const int jc = j > 0 ? j : 0;
        const float w0 = w[jc], w1 = w[jc >= 1 ? jc - 1 : 0], w2 = w[jc >= 2 ? jc - 2 : 0], w3 = w[jc >= 3 ? jc - 3 : 0];
        acc = fmaf(s_x[q], lwo[i], acc);
        acc = fmaf(j >= 0 ? v.x : 0.f, w0, acc);
        acc = fmaf(j >= 1 ? v.y : 0.f, w1, acc);
        acc = fmaf(j >= 2 ? v.z : 0.f, w2, acc);
        acc = fmaf(j >= 3 ? v.w : 0.f, w3, acc);
      }
    }
    if (nsplit > 1) {
      __syncthreads();
      s_part[tid] = acc;
      __syncthreads();
      if (live && part == 0 && b < B) {
        float t = lb[o];
        for (int p = 0; p < nsplit; ++p) t += s_part[p * items + e];
        out[(size_t)b * out_f + o] = act_apply(t, act);
      }
    } else if (live && b < B) {
      out[(size_t)b * out_f + o] = act_apply(acc + lb[o], act);
    }
  }
}

// backward wrt the layer input: thread = (sample, feature)
__global__ __launch_bounds__(1024) void kan_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ knots, const float* __restrict__ lw,
                                                         const float* __restrict__ y, const float* __restrict__ gy,
                                                         float* __restrict__ dx, int B, int in_f, int out_f, int nk,
                                                         int TB, int act, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_knots = smem;
  float* s_g = s_knots + KAN_MAX_KNOTS;      // TB * out_f : dL/dz
  const int tid = threadIdx.x, T = blockDim.x;
  const int nb = nk - 4;
  const int b0 = blockIdx.x * TB;
  if (tid < nk) s_knots[tid] = knots[tid];
  for (int e = tid; e < TB * out_f; e += T) {
    const int bl = e / out_f, b = b0 + bl;
    s_g[e] = b < B ? act_grad(gy[(size_t)b0 * out_f + e], y[(size_t)b0 * out_f + e], act) : 0.f;
  }
  __syncthreads();
  const float inv_h0 = 1.f / (s_knots[1] - s_knots[0]);
  // four adjacent lanes share one (sample, feature): each walks a quarter of the outputs (a chain of L2 gathers whose
  // length sets the time), the partial sums meet through two cross-lane adds
  const int part = tid & 3;
  for (int e0 = 0; e0 < TB * in_f; e0 += T / 4) {
    const int e = e0 + (tid >> 2);
    const bool live = e < TB * in_f && b0 + e / in_f < B;
    const int ec = live ? e : 0;
    const int bl = ec / in_f, i = ec - bl * in_f, b = b0 + bl;
    const float xv = x[(size_t)(b < B ? b : B - 1) * in_f + i];
    const float xn = tanhf(xv);
    float dv[4];
    const Basis4 bs = kan_basis<true>(xn, s_knots, nk, inv_h0, dv);
    const float* g = s_g + bl * out_f;
    float lin = 0.f, spl = 0.f;
    const int jc = bs.j > 0 ? bs.j : 0;
    const int j1 = jc >= 1 ? jc - 1 : 0, j2 = jc >= 2 ? jc - 2 : 0, j3 = jc >= 3 ? jc - 3 : 0;
    const float d0 = bs.j >= 0 ? dv[0] : 0.f, d1 = bs.j >= 1 ? dv[1] : 0.f, d2 = bs.j >= 2 ? dv[2] : 0.f, d3 = bs.j >= 3 ? dv[3] : 0.f;
#pragma unroll 8
    for (int o = part; o < out_f; o += 4) {               // branch-free: loads of several outputs in flight
      const float go = g[o];
      const float* w = W + ((size_t)i * out_f + o) * nb;
      const float w0 = w[jc], w1 = w[j1], w2 = w[j2], w3 = w[j3];
      lin = fmaf(go, lw[(size_t)o * in_f + i], lin);
      float sv = d0 * w0;
      sv = fmaf(d1, w1, sv);
      sv = fmaf(d2, w2, sv);
      sv = fmaf(d3, w3, sv);
      spl = fmaf(go, sv, spl);
    }
    lin += __shfl_xor(lin, 1); spl += __shfl_xor(spl, 1);
    lin += __shfl_xor(lin, 2); spl += __shfl_xor(spl, 2);
    if (live && part == 0) {
      const float r = fmaf(spl, 1.f - xn * xn, lin);       // d tanh; clamp is the identity on (-1, 1)
      float* p = dx + (size_t)b * in_f + i;
      *p = accumulate ? *p + r : r;
    }
  }
}

// backward wrt the parameters: one workgroup = one input feature i (owns dW[i,:,:] and dlin_w[:,i]).
__global__ __launch_bounds__(1024) void kan_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ knots,
                                                         const float* __restrict__ y, const float* __restrict__ gy,
                                                         float* __restrict__ dW, float* __restrict__ dlw,
                                                         float* __restrict__ dlb, int B, int in_f, int out_f, int nk,
                                                         int BC, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_knots = smem;
  float* s_x = s_knots + KAN_MAX_KNOTS;      // BC
  float* s_d = s_x + BC;                     // BC * nb dense basis of feature i
  float* s_gz = s_d + BC * (nk - 4);         // BC * out_f  dL/dz of this batch chunk (read many times below)
  float* s_part = s_gz + BC * out_f;         // blockDim partial sums
  const int tid = threadIdx.x, T = blockDim.x;
  const int nb = nk - 4;
  const int i = blockIdx.x;
  if (tid < nk) s_knots[tid] = knots[tid];
  __syncthreads();
  const float inv_h0 = 1.f / (s_knots[1] - s_knots[0]);
  const int n_sp = out_f * nb;
  const int n_items = n_sp + out_f + (i == 0 ? out_f : 0);
  // The sample loop of one item is a serial chain (2 LDS reads + 1 FMA per sample), so when there are fewer items
  // than threads each item is shared by `nsplit` threads that take a slice of the samples (layer 3 has 9 items).
  const int nsplit = n_items >= T ? 1 : T / n_items;
  for (int c0 = 0; c0 < B; c0 += BC) {
    const int nbatch = min(BC, B - c0);
    __syncthreads();
    for (int bl = tid; bl < nbatch; bl += T) {
      const float xv = x[(size_t)(c0 + bl) * in_f + i];
      const Basis4 bs = kan_basis<false>(tanhf(xv), s_knots, nk, inv_h0, nullptr);
      float* row = s_d + bl * nb;
      for (int k = 0; k < nb; ++k) row[k] = 0.f;
      if (bs.j >= 0) {
        row[bs.j] = bs.v[0];
        if (bs.j >= 1) row[bs.j - 1] = bs.v[1];
        if (bs.j >= 2) row[bs.j - 2] = bs.v[2];
        if (bs.j >= 3) row[bs.j - 3] = bs.v[3];
      }
      s_x[bl] = xv;
    }
    {
      // dL/dz of the chunk: issue every load of a thread before the first use (latency-, not bandwidth-bound)
      const int n_e = nbatch * out_f;
      const size_t q0 = (size_t)c0 * out_f;
      for (int e0 = tid; e0 < n_e; e0 += T * 8) {
        float gv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * T;
          const int ec = e < n_e ? e : n_e - 1;
          gv[u] = gy[q0 + ec]; yv[u] = y[q0 + ec];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * T;
          if (e < n_e) s_gz[e] = act_grad(gv[u], yv[u], act);
        }
      }
    }
    __syncthreads();
    for (int e0 = 0; e0 < n_items; e0 += T) {
      const int e = e0 + (nsplit == 1 ? tid : tid % n_items);
      const int part = nsplit == 1 ? 0 : tid / n_items;
      const bool live = e < n_items && part < nsplit;
      int o = 0, k = 0, kind = 2;             // kind 0: spline weight, 1: linear weight, 2: linear bias
      if (live) {
        if (e < n_sp) { kind = 0; k = e / out_f; o = e - k * out_f; }
        else if (e < n_sp + out_f) { kind = 1; o = e - n_sp; }
        else { kind = 2; o = e - n_sp - out_f; }
      }
      float acc = 0.f;
      if (live) {
        const int chunk = (nbatch + nsplit - 1) / nsplit;
        const int lo = part * chunk, hi = min(nbatch, lo + chunk);
        const float* mp = kind == 0 ? s_d + k : s_x;          // multiplier stream: stride nb (spline) or 1 (linear)
        const int ms = kind == 0 ? nb : 1;
        if (kind == 2) {
#pragma unroll 8
          for (int bl = lo; bl < hi; ++bl) acc += s_gz[bl * out_f + o];
        } else {
#pragma unroll 8
          for (int bl = lo; bl < hi; ++bl) acc = fmaf(s_gz[bl * out_f + o], mp[bl * ms], acc);
        }
      }
      if (nsplit > 1) {
        s_part[tid] = acc;
        __syncthreads();
        if (live && part == 0) {
          acc = 0.f;
          for (int p = 0; p < nsplit; ++p) acc += s_part[p * n_items + e];
        }
        __syncthreads();
      }
      if (live && part == 0) {
        float* p = kind == 0 ? dW + ((size_t)i * out_f + o) * nb + k : (kind == 1 ? dlw + (size_t)o * in_f + i : dlb + o);
        *p = c0 == 0 ? acc : *p + acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small fp32 linear layers for the heads (192 -> 128 -> {4,3,1,1}); B is a few hundred rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      float* __restrict__ y, int B, int in_f, int out_f, int flags) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * out_f) return;
  const int b = e / out_f, o = e - b * out_f;
  const float4* xr = (const float4*)(x + (size_t)b * in_f);
  const float4* wr = (const float4*)(w + (size_t)o * in_f);
  float acc = bias ? bias[o] : 0.f;
  for (int i = 0; i < in_f / 4; ++i) {
    const float4 a = xr[i], c = wr[i];
    acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
  }
  if (flags & ROVIT_LIN_RELU) acc = fmaxf(acc, 0.f);
  if (mask) acc *= mask[e];
  if (flags & ROVIT_LIN_CLAMP10) acc = fminf(fmaxf(acc, -10.f), 10.f);       // heads.py:100
  y[e] = acc;
}

__device__ __forceinline__ float clamp_gate(float g, const float* yc, size_t idx) {
  if (!yc) return g;
  const float v = yc[idx];
  return (v > -10.f && v < 10.f) ? g : 0.f;
}

// dx[b,i] (+)= (sum_o g[b,o] w[o,i]) * mul[b,i] * [pos[b,i] > 0]
__global__ __launch_bounds__(256) void lin_bwd_dx_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                         const float* __restrict__ yclamp, const float* __restrict__ mul,
                                                         const float* __restrict__ pos, float* __restrict__ dx, int B,
                                                         int in_f, int out_f, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * in_f) return;
  const int b = e / in_f, i = e - b * in_f;
  float acc = 0.f;
#pragma unroll 8
  for (int o = 0; o < out_f; ++o)
    acc = fmaf(clamp_gate(g[(size_t)b * out_f + o], yclamp, (size_t)b * out_f + o), w[(size_t)o * in_f + i], acc);
  if (mul) acc *= mul[e];
  if (pos) acc = pos[e] > 0.f ? acc : 0.f;
  dx[e] = accumulate ? dx[e] + acc : acc;
}

// dw[o,i] = sum_b g[b,o] x[b,i];  db[o] = sum_b g[b,o]
__global__ __launch_bounds__(256) void lin_bwd_dw_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                         const float* __restrict__ yclamp, float* __restrict__ dw,
                                                         float* __restrict__ db, int B, int in_f, int out_f) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= out_f * in_f) return;
  const int o = e / in_f, i = e - o * in_f;
  float acc = 0.f, accb = 0.f;
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    const float gv = clamp_gate(g[(size_t)b * out_f + o], yclamp, (size_t)b * out_f + o);
    acc = fmaf(gv, x[(size_t)b * in_f + i], acc);
    accb += gv;
  }
  dw[e] = acc;
  if (i == 0) db[o] = accb;
}

// dense (n, nb) basis table: the standalone form of BSplineBasis.compute_basis (models/kan.py:8-44); the input is
// already the normalised coordinate (no tanh), as in KANLayer.plot_activation (kan.py:100-114).
__global__ __launch_bounds__(256) void kan_basis_kernel(const float* __restrict__ xn, const float* __restrict__ knots,
                                                        float* __restrict__ out, int n, int nk) {
  __shared__ float s_knots[KAN_MAX_KNOTS];
  if (threadIdx.x < nk) s_knots[threadIdx.x] = knots[threadIdx.x];
  __syncthreads();
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int nb = nk - 4;
  const Basis4 bs = kan_basis<false>(xn[e], s_knots, nk, 1.f / (s_knots[1] - s_knots[0]), nullptr);
  float* row = out + (size_t)e * nb;
  for (int k = 0; k < nb; ++k) row[k] = 0.f;
  if (bs.j >= 0) {
    row[bs.j] = bs.v[0];
    if (bs.j >= 1) row[bs.j - 1] = bs.v[1];
    if (bs.j >= 2) row[bs.j - 2] = bs.v[2];
    if (bs.j >= 3) row[bs.j - 3] = bs.v[3];
  }
}

// ---- batched variants: the heads are 3 + 4 tiny linears forward and 7 backward problems; running each as its own
// launch costs more in launch gaps than in work, so they go out as 2 + 4 launches with the problem list passed by
// value in the kernel arguments.
struct LinFwdDesc { const float* x; const float* w; const float* bias; const float* mask; float* y; int in_f, out_f, flags; };
struct LinFwdBatch { LinFwdDesc d[4]; int first[5]; int n, B; };

__global__ __launch_bounds__(256) void lin_fwd_batch_kernel(const LinFwdBatch pb) {
  // four lanes per output element, each walking a quarter of the input features with 16-byte loads (in_f = 192: twelve
  // loads per lane, all in flight together), combined with two cross-lane adds: the dot product is a latency chain, not work
  int i = 0;
  while (i + 1 < pb.n && (int)blockIdx.x >= pb.first[i + 1]) ++i;
  const LinFwdDesc& d = pb.d[i];
  const int t = ((int)blockIdx.x - pb.first[i]) * 256 + threadIdx.x;
  const int e = t >> 2, part = t & 3;
  const bool live = e < pb.B * d.out_f;
  const int ec = live ? e : 0;
  const int b = ec / d.out_f, o = ec - b * d.out_f;
  const float4* xr = (const float4*)(d.x + (size_t)b * d.in_f);
  const float4* wr = (const float4*)(d.w + (size_t)o * d.in_f);
  const int n4 = d.in_f / 4;
  float acc = 0.f;
#pragma unroll 12
  for (int k = part; k < n4; k += 4) {
    const float4 a = xr[k], c = wr[k];
    acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  if (!live || part != 0) return;
  if (d.bias) acc += d.bias[o];
  if (d.flags & ROVIT_LIN_RELU) acc = fmaxf(acc, 0.f);
  if (d.mask) acc *= d.mask[e];
  if (d.flags & ROVIT_LIN_CLAMP10) acc = fminf(fmaxf(acc, -10.f), 10.f);
  d.y[e] = acc;
}

// dx[b,i] (+)= (sum over sources s, outputs o of g_s[b,o] w_s[o,i]) * mul[b,i] * [pos[b,i] > 0]
struct LinDxDesc { int nsrc; const float* g[3]; const float* w[3]; const float* yc[3]; int out_f[3];
                   const float* mul; const float* pos; float* dx; int in_f, accumulate; };
struct LinDxBatch { LinDxDesc d[3]; int first[4]; int n, B; };

__global__ __launch_bounds__(256) void lin_bwd_dx_batch_kernel(const LinDxBatch pb) {
  int i = 0;
  while (i + 1 < pb.n && (int)blockIdx.x >= pb.first[i + 1]) ++i;
  const LinDxDesc& d = pb.d[i];
  const int e = ((int)blockIdx.x - pb.first[i]) * 256 + threadIdx.x;
  if (e >= pb.B * d.in_f) return;
  const int b = e / d.in_f, k = e - b * d.in_f;
  float acc = 0.f;
  for (int s = 0; s < d.nsrc; ++s) {
#pragma unroll 16
    for (int o = 0; o < d.out_f[s]; ++o) {
      const size_t q = (size_t)b * d.out_f[s] + o;
      acc = fmaf(clamp_gate(d.g[s][q], d.yc[s], q), d.w[s][(size_t)o * d.in_f + k], acc);
    }
  }
  if (d.mul) acc *= d.mul[e];
  if (d.pos) acc = d.pos[e] > 0.f ? acc : 0.f;
  d.dx[e] = d.accumulate ? d.dx[e] + acc : acc;
}

struct LinDwDesc { const float* g; const float* x; const float* yc; float* dw; float* db; int in_f, out_f; };
struct LinDwBatch { LinDwDesc d[8]; int first[9]; int n, B; };

// One workgroup = 64 consecutive dW elements x 4 batch slices (wave w sums samples w, w+4, ...), partial sums
// combined through LDS: the sample loop is a chain of dependent global loads, so its length, not the arithmetic, sets
// the kernel time (62 us with one thread walking all 256 samples).
__device__ __forceinline__ void lin_dw_body(const LinDwBatch pb, int bid) {
  __shared__ float s_w[16][64], s_b[16][64];
  int i = 0;
  while (i + 1 < pb.n && bid >= pb.first[i + 1]) ++i;
  const LinDwDesc& d = pb.d[i];
  const int el = threadIdx.x & 63, bs = threadIdx.x >> 6;          // 16 batch slices: 16 samples per thread at batch 256
  const int e = (bid - pb.first[i]) * 64 + el;
  const bool live = e < d.out_f * d.in_f;
  const int o = live ? e / d.in_f : 0, k = live ? e - o * d.in_f : 0;
  float acc = 0.f, accb = 0.f;
#pragma unroll 16
  for (int b = bs; b < pb.B; b += 16) {
    const size_t q = (size_t)b * d.out_f + o;
    const float gv = clamp_gate(d.g[q], d.yc, q);
    acc = fmaf(gv, d.x[(size_t)b * d.in_f + k], acc);
    accb += gv;
  }
  s_w[bs][el] = acc; s_b[bs][el] = accb;
  __syncthreads();
  if (bs == 0 && live) {
    float tw = 0.f, tb = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { tw += s_w[q][el]; tb += s_b[q][el]; }
    d.dw[e] = tw;
    if (k == 0) d.db[o] = tb;
  }
}
__global__ __launch_bounds__(1024) void lin_bwd_dw_batch_kernel(const LinDwBatch pb) { lin_dw_body(pb, (int)blockIdx.x); }

template <class Batch, class Kernel>
int launch_lin_batch(Batch& pb, int n, const int* work, Kernel kern, const char* name, hipStream_t st, int per_block = 256, int threads = 256) {
  int blocks = 0;
  for (int i = 0; i < n; ++i) { pb.first[i] = blocks; blocks += (work[i] + per_block - 1) / per_block; }
  pb.first[n] = blocks; pb.n = n;
  if (blocks == 0) return ROVIT_OK;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, st, pb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { rovit_set_error("%s: launch failed: %s", name, hipGetErrorString(e)); return ROVIT_ERR_LAUNCH; }
  return ROVIT_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the whole KAN stack in TWO launches (round 3; six before: a dx and a dW launch per layer).
//   1. kan_stack_bwd_dx_kernel: the chain dL/dz_3 -> dx_3 -> dL/dz_2 -> ... -> dx_1 is local to a sample, so one workgroup
//      walks all layers for its TB samples with the inter-layer gradients in LDS; it writes dL/dz of every layer (what the
//      parameter gradients need) and the gradient w.r.t. the stack input.  Per layer the arithmetic and the summation order
//      are those of kan_bwd_dx_kernel (four lanes share a (sample, feature) pair).
//   2. kan_stack_bwd_dw_kernel: one workgroup per (layer, input feature) = 192 + 64 + 16 workgroups in one launch; the body of
//      kan_bwd_dw_kernel reading dL/dz directly.
// Reference: autograd of KANSeverityModule.forward (/root/reference/models/kan.py:138-149, KANLayer.forward :70-95) as
// reached from training/trainer.py:119,136.
// ---------------------------------------------------------------------------------------------
constexpr int KB_MAX_LAYERS = 4;
constexpr int KB_MAXW = 64;                 // widest layer output (and input of every layer but the first)
struct KanStackBwdArgs {
  const float* x; int B; int nl;
  int dims[KB_MAX_LAYERS + 1];
  int nk[KB_MAX_LAYERS];
  int act[KB_MAX_LAYERS];
  const float* W[KB_MAX_LAYERS];            // spline weights (in, out, nb): reference layout
  const float* knots[KB_MAX_LAYERS];
  const float* lw[KB_MAX_LAYERS];           // linear weight (out, in)
  const float* y[KB_MAX_LAYERS];            // post-activation outputs (B, out)
  const float* gy[KB_MAX_LAYERS];           // gradient w.r.t. each layer's output coming from OUTSIDE the stack (NULL: none)
  float* gz[KB_MAX_LAYERS];                 // (B, out) dL/dz, written by the dx kernel, read by the dW kernel
  float* dx;                                // (B, in_0) or NULL
  float* dW[KB_MAX_LAYERS]; float* dlw[KB_MAX_LAYERS]; float* dlb[KB_MAX_LAYERS];
  int bc[KB_MAX_LAYERS];                    // batch rows per LDS chunk of the dW kernel
  int thr[KB_MAX_LAYERS];                   // working threads of a layer's dW workgroups (as rovit_kan_layer_bwd picks them)
  int wg0[KB_MAX_LAYERS + 1];               // first workgroup of each layer in the dW launch
};

template <int L>
__device__ __forceinline__ void kb_dx_layer(const KanStackBwdArgs& a, float* s_knots, float* s_gz, float* s_gin, float* s_gout, int b0,
                                            int TB, int tid, int T) {
  const int in_f = a.dims[L], out_f = a.dims[L + 1], nk = a.nk[L], nb = nk - 4;
  __syncthreads();                                     // previous layer done with s_knots / s_gz; its s_gout (our s_gin) complete
  if (tid < nk) s_knots[tid] = a.knots[L][tid];
  for (int e = tid; e < TB * out_f; e += T) {
    const int bl = e / out_f, o = e - bl * out_f, b = b0 + bl;
    float gz = 0.f;
    if (b < a.B) {
      float g = (L + 1 < KB_MAX_LAYERS && L + 1 < a.nl) ? s_gin[bl * KB_MAXW + o] : 0.f;       // from the layer above (its dx)
      if (a.gy[L]) g += a.gy[L][(size_t)b * out_f + o];
      gz = act_grad(g, a.y[L][(size_t)b * out_f + o], a.act[L]);
      a.gz[L][(size_t)b * out_f + o] = gz;
    }
    s_gz[e] = gz;
  }
  __syncthreads();
  if (L == 0 && !a.dx) return;
  const float* x = L == 0 ? a.x : a.y[L - 1];           // this layer's input
  const float* W = a.W[L];
  const float* lw = a.lw[L];
  const float inv_h0 = 1.f / (s_knots[1] - s_knots[0]);
  const int part = tid & 3;
  for (int e0 = 0; e0 < TB * in_f; e0 += T / 4) {
    const int e = e0 + (tid >> 2);
    const bool live = e < TB * in_f && b0 + e / in_f < a.B;
    const int ec = live ? e : 0;
    const int bl = ec / in_f, i = ec - bl * in_f, b = b0 + bl;
    const float xv = x[(size_t)(b < a.B ? b : a.B - 1) * in_f + i];
    const float xn = tanhf(xv);
    float dv[4];
    const Basis4 bs = kan_basis<true>(xn, s_knots, nk, inv_h0, dv);
    const float* g = s_gz + bl * out_f;
    float lin = 0.f, spl = 0.f;
    const int jc = bs.j > 0 ? bs.j : 0;
    const int j1 = jc >= 1 ? jc - 1 : 0, j2 = jc >= 2 ? jc - 2 : 0, j3 = jc >= 3 ? jc - 3 : 0;
    const float d0 = bs.j >= 0 ? dv[0] : 0.f, d1 = bs.j >= 1 ? dv[1] : 0.f, d2 = bs.j >= 2 ? dv[2] : 0.f, d3 = bs.j >= 3 ? dv[3] : 0.f;
#pragma unroll 8
    for (int o = part; o < out_f; o += 4) {               // branch-free: loads of several outputs in flight
      const float go = g[o];
      const float* w = W + ((size_t)i * out_f + o) * nb;
      const float w0 = w[jc], w1 = w[j1], w2 = w[j2], w3 = w[j3];
      lin = fmaf(go, lw[(size_t)o * in_f + i], lin);
      float sv = d0 * w0;
      sv = fmaf(d1, w1, sv);
      sv = fmaf(d2, w2, sv);
      sv = fmaf(d3, w3, sv);
      spl = fmaf(go, sv, spl);
    }
    lin += __shfl_xor(lin, 1); spl += __shfl_xor(spl, 1);
    lin += __shfl_xor(lin, 2); spl += __shfl_xor(spl, 2);
    if (live && part == 0) {
      const float r = fmaf(spl, 1.f - xn * xn, lin);       // d tanh; clamp is the identity on (-1, 1)
      if (L == 0) a.dx[(size_t)b * in_f + i] = r;
      else s_gout[bl * KB_MAXW + i] = r;                   // gradient w.r.t. the output of layer L-1
    }
  }
}

__global__ __launch_bounds__(1024) void kan_stack_bwd_dx_kernel(const KanStackBwdArgs a, int TB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_knots = smem;                                 // [64]
  float* s_gz = s_knots + KAN_MAX_KNOTS;                 // [TB][<= 64] dL/dz of the current layer
  float* s_g0 = s_gz + TB * KB_MAXW;                     // inter-layer gradients, ping-pong
  float* s_g1 = s_g0 + TB * KB_MAXW;
  const int tid = threadIdx.x, T = blockDim.x;
  const int b0 = blockIdx.x * TB;
  // layer l reads the gradient of its output from buffer (l & 1) and writes the gradient of its input to ((l - 1) & 1)
  if (a.nl > 3) kb_dx_layer<3>(a, s_knots, s_gz, s_g1, s_g0, b0, TB, tid, T);
  if (a.nl > 2) kb_dx_layer<2>(a, s_knots, s_gz, s_g0, s_g1, b0, TB, tid, T);
  if (a.nl > 1) kb_dx_layer<1>(a, s_knots, s_gz, s_g1, s_g0, b0, TB, tid, T);
  kb_dx_layer<0>(a, s_knots, s_gz, s_g0, s_g1, b0, TB, tid, T);
}

// parameter gradients of every layer: workgroup -> (layer, input feature i); owns dW[i,:,:], dlin_w[:,i] (and dlin_b for i = 0)
__device__ __forceinline__ void kan_dw_body(const KanStackBwdArgs a, const int bid, float* smem) {
  int l = 0;
#pragma unroll
  for (int q = 1; q < KB_MAX_LAYERS; ++q)
    if (q < a.nl && bid >= a.wg0[q]) l = q;
  // (scalar copies of the selected layer: l is workgroup-uniform)
  int in_f = a.dims[0], out_f = a.dims[1], nk = a.nk[0], BC = a.bc[0], wg0 = a.wg0[0], T = a.thr[0];
  const float *x = a.x, *knots = a.knots[0], *gzp = a.gz[0];
  float *dW = a.dW[0], *dlw = a.dlw[0], *dlb = a.dlb[0];
#pragma unroll
  for (int q = 1; q < KB_MAX_LAYERS; ++q)
    if (l == q) {
      in_f = a.dims[q]; out_f = a.dims[q + 1]; nk = a.nk[q]; BC = a.bc[q]; wg0 = a.wg0[q]; T = a.thr[q];
      x = a.y[q - 1]; knots = a.knots[q]; gzp = a.gz[q]; dW = a.dW[q]; dlw = a.dlw[q]; dlb = a.dlb[q];
    }
  const int B = a.B;
  float* s_knots = smem;
  float* s_x = s_knots + KAN_MAX_KNOTS;      // BC
  float* s_d = s_x + BC;                     // BC * nb dense basis of feature i
  float* s_gz = s_d + BC * (nk - 4);         // BC * out_f  dL/dz of this batch chunk (read many times below)
  float* s_part = s_gz + BC * out_f;         // blockDim partial sums
  // T = the thread count rovit_kan_layer_bwd gives this layer's kan_bwd_dw_kernel (256 or 1024): the same split of the sample
  // sums, hence bit-identical gradients; threads beyond T only take part in the barriers
  const int tid = threadIdx.x;
  const bool on = tid < T;
  const int nb = nk - 4;
  const int i = bid - wg0;
  if (tid < nk) s_knots[tid] = knots[tid];
  __syncthreads();
  const float inv_h0 = 1.f / (s_knots[1] - s_knots[0]);
  const int n_sp = out_f * nb;
  const int n_items = n_sp + out_f + (i == 0 ? out_f : 0);
  const int nsplit = n_items >= T ? 1 : T / n_items;
  for (int c0 = 0; c0 < B; c0 += BC) {
    const int nbatch = min(BC, B - c0);
    __syncthreads();
    for (int bl = on ? tid : nbatch; bl < nbatch; bl += T) {
      const float xv = x[(size_t)(c0 + bl) * in_f + i];
      const Basis4 bs = kan_basis<false>(tanhf(xv), s_knots, nk, inv_h0, nullptr);
      float* row = s_d + bl * nb;
      for (int k = 0; k < nb; ++k) row[k] = 0.f;
      if (bs.j >= 0) {
        row[bs.j] = bs.v[0];
        if (bs.j >= 1) row[bs.j - 1] = bs.v[1];
        if (bs.j >= 2) row[bs.j - 2] = bs.v[2];
        if (bs.j >= 3) row[bs.j - 3] = bs.v[3];
      }
      s_x[bl] = xv;
    }
    {
      const int n_e = nbatch * out_f;
      const size_t q0 = (size_t)c0 * out_f;
      for (int e0 = on ? tid : n_e; e0 < n_e; e0 += T * 8) {
        float gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * T;
          gv[u] = gzp[q0 + (e < n_e ? e : n_e - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * T;
          if (e < n_e) s_gz[e] = gv[u];
        }
      }
    }
    __syncthreads();
    for (int e0 = 0; e0 < n_items; e0 += T) {
      const int e = e0 + (nsplit == 1 ? tid : tid % n_items);
      const int part = nsplit == 1 ? 0 : tid / n_items;
      const bool live = on && e < n_items && part < nsplit;
      int o = 0, k = 0, kind = 2;             // kind 0: spline weight, 1: linear weight, 2: linear bias
      if (live) {
        if (e < n_sp) { kind = 0; k = e / out_f; o = e - k * out_f; }
        else if (e < n_sp + out_f) { kind = 1; o = e - n_sp; }
        else { kind = 2; o = e - n_sp - out_f; }
      }
      float acc = 0.f;
      if (live) {
        const int chunk = (nbatch + nsplit - 1) / nsplit;
        const int lo = part * chunk, hi = min(nbatch, lo + chunk);
        const float* mp = kind == 0 ? s_d + k : s_x;          // multiplier stream: stride nb (spline) or 1 (linear)
        const int ms = kind == 0 ? nb : 1;
        if (kind == 2) {
#pragma unroll 8
          for (int bl = lo; bl < hi; ++bl) acc += s_gz[bl * out_f + o];
        } else {
#pragma unroll 8
          for (int bl = lo; bl < hi; ++bl) acc = fmaf(s_gz[bl * out_f + o], mp[bl * ms], acc);
        }
      }
      if (nsplit > 1) {
        if (on) s_part[tid] = acc;
        __syncthreads();
        if (live && part == 0) {
          acc = 0.f;
          for (int p = 0; p < nsplit; ++p) acc += s_part[p * n_items + e];
        }
        __syncthreads();
      }
      if (live && part == 0) {
        float* p = kind == 0 ? dW + ((size_t)i * out_f + o) * nb + k : (kind == 1 ? dlw + (size_t)o * in_f + i : dlb + o);
        *p = c0 == 0 ? acc : *p + acc;
      }
    }
  }
}

__global__ __launch_bounds__(1024) void kan_stack_bwd_dw_kernel(const KanStackBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kan_dw_body(a, (int)blockIdx.x, smem);
}

// every parameter gradient of the head phase in ONE grid (head_phase.hip): workgroups [0, kan_wgs) are the (layer, input feature)
// workgroups of the KAN stack, the rest the 64-element slices of the head linears' weight gradients
__global__ __launch_bounds__(1024) void head_phase_dw_kernel(const KanStackBwdArgs a, const LinDwBatch pb, int kan_wgs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < kan_wgs) kan_dw_body(a, (int)blockIdx.x, smem);
  else lin_dw_body(pb, (int)blockIdx.x - kan_wgs);
}

int kan_tb(int out_f) { int tb = 64 / (out_f > 0 ? out_f : 1); return tb < 1 ? 1 : (tb > 16 ? 16 : tb); }

}  // namespace

extern "C" int rovit_kan_basis(const float* x_norm, const float* knots, float* basis, int n, int n_knots, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x_norm && knots && basis, ROVIT_ERR_NULL, "kan_basis: null pointer");
  ROVIT_CHECK_ARG(n > 0 && n_knots >= 8 && n_knots <= KAN_MAX_KNOTS, ROVIT_ERR_SHAPE, "kan_basis: bad shape n=%d knots=%d", n, n_knots);
  hipLaunchKernelGGL(kan_basis_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x_norm, knots, basis, n, n_knots);
  ROVIT_CHECK_LAUNCH("kan_basis_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_kan_layer_fwd(const float* x, const float* spline_w, const float* knots, const float* lin_w,
                                   const float* lin_b, float* out, int batch, int in_f, int out_f, int n_knots, int act,
                                   rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && spline_w && knots && lin_w && lin_b && out, ROVIT_ERR_NULL, "kan_layer_fwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && in_f > 0 && out_f > 0, ROVIT_ERR_SHAPE, "kan_layer_fwd: bad shape B=%d in=%d out=%d", batch, in_f, out_f);
  ROVIT_CHECK_ARG(n_knots >= 8 && n_knots <= KAN_MAX_KNOTS, ROVIT_ERR_SHAPE,
                  "kan_layer_fwd: degree-3 layer needs 8..%d knots, got %d", KAN_MAX_KNOTS, n_knots);
  const int tb = kan_tb(out_f);
  // more feature slices per (sample, output) item -- for short basis rows only: at num_knots 32 (136-byte rows, one cache line
  // per gathered (feature, output) pair) four times the threads only add L2 pressure (C5: 51 -> 84 us, measured)
  // ... and for batches that do not fill the chip with 256-thread workgroups anyway (batch 65536: 2.35 -> 3.33 ms with 1024)
  const int threads = (tb * out_f <= 64 && in_f >= 32 && n_knots <= 16 && batch <= 1024) ? 1024 : 256;
  const size_t lds = (KAN_MAX_KNOTS + (size_t)tb * in_f * 6 + threads) * sizeof(float);
  ROVIT_CHECK_ARG(lds <= 64 * 1024, ROVIT_ERR_SHAPE, "kan_layer_fwd: in_features %d too large for the LDS tile", in_f);
  hipLaunchKernelGGL(kan_fwd_kernel, dim3((batch + tb - 1) / tb), dim3(threads), lds, (hipStream_t)stream, x, spline_w, knots,
                     lin_w, lin_b, out, batch, in_f, out_f, n_knots, tb, act);
  ROVIT_CHECK_LAUNCH("kan_fwd_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_kan_layer_bwd(const float* x, const float* spline_w, const float* knots, const float* lin_w,
                                   const float* out, const float* grad_out, float* dx, float* d_spline_w, float* d_lin_w,
                                   float* d_lin_b, int batch, int in_f, int out_f, int n_knots, int act, int accumulate_dx,
                                   rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && spline_w && knots && lin_w && out && grad_out, ROVIT_ERR_NULL, "kan_layer_bwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && in_f > 0 && out_f > 0, ROVIT_ERR_SHAPE, "kan_layer_bwd: bad shape");
  ROVIT_CHECK_ARG(n_knots >= 8 && n_knots <= KAN_MAX_KNOTS, ROVIT_ERR_SHAPE, "kan_layer_bwd: bad knot count %d", n_knots);
  if (dx) {
    const int tb = kan_tb(out_f);
    const size_t lds = (KAN_MAX_KNOTS + (size_t)tb * out_f) * sizeof(float);
    hipLaunchKernelGGL(kan_bwd_dx_kernel, dim3((batch + tb - 1) / tb), dim3(tb * in_f * 4 >= 1024 ? 1024 : 256), lds, (hipStream_t)stream, x, spline_w,
                       knots, lin_w, out, grad_out, dx, batch, in_f, out_f, n_knots, tb, act, accumulate_dx);
    ROVIT_CHECK_LAUNCH("kan_bwd_dx_kernel");
  }
  if (d_spline_w) {
    ROVIT_CHECK_ARG(d_lin_w && d_lin_b, ROVIT_ERR_NULL, "kan_layer_bwd: parameter gradients must be given together");
    const int nb = n_knots - 4;
    int bc = (24 * 1024) / (nb + 1 + out_f);   // batch rows whose basis + dL/dz stay in LDS (<= 96 KB)
    bc = bc > batch ? batch : bc;
    const int n_items = out_f * (nb + 2);
    const int threads = n_items >= 512 ? 1024 : 256;         // one or more threads per (output, basis) item
    const size_t lds = (KAN_MAX_KNOTS + (size_t)bc * (nb + 1 + out_f) + threads) * sizeof(float);
    (void)rovit_set_max_lds((const void*)kan_bwd_dw_kernel, (size_t)(104 * 1024));
    hipLaunchKernelGGL(kan_bwd_dw_kernel, dim3(in_f), dim3(threads), lds, (hipStream_t)stream, x, knots, out, grad_out,
                       d_spline_w, d_lin_w, d_lin_b, batch, in_f, out_f, n_knots, bc, act);
    ROVIT_CHECK_LAUNCH("kan_bwd_dw_kernel");
  }
  return ROVIT_OK;
}

// Backward of the whole stack in two launches (see kan_stack_bwd_dx_kernel).  Host arrays of n_layers device pointers:
// spline_w (in,out,nb) / knots / lin_w (out,in): the layers' parameters (reference layouts); outs: post-activation outputs of
// the forward; grad_outs: gradient w.r.t. each layer's output from outside the stack (NULL entries allowed; normally only
// the last is set); gz: scratch (batch, out_l) per layer; d_spline_w / d_lin_w / d_lin_b: parameter gradients (all set, or
// d_spline_w == NULL for none); dx: gradient w.r.t. the stack input (batch, dims[0]) or NULL.
extern "C" int rovit_kan_stack_bwd(const float* x, const float* const* spline_w, const float* const* knots, const float* const* lin_w,
                                   const float* const* outs, const float* const* grad_outs, float* const* gz, float* dx,
                                   float* const* d_spline_w, float* const* d_lin_w, float* const* d_lin_b, int batch, const int* dims,
                                   const int* n_knots, const int* acts, int n_layers, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && spline_w && knots && lin_w && outs && grad_outs && gz && dims && n_knots && acts, ROVIT_ERR_NULL, "kan_stack_bwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && n_layers >= 1 && n_layers <= KB_MAX_LAYERS, ROVIT_ERR_SHAPE, "kan_stack_bwd: 1..%d layers", KB_MAX_LAYERS);
  KanStackBwdArgs a{};
  a.x = x; a.B = batch; a.nl = n_layers; a.dx = dx;
  for (int l = 0; l <= n_layers; ++l) a.dims[l] = dims[l];
  const bool want_dw = d_spline_w != nullptr;
  size_t lds_dw = 0;
  int wgs = 0;
  for (int l = 0; l < n_layers; ++l) {
    ROVIT_CHECK_ARG(spline_w[l] && knots[l] && lin_w[l] && outs[l] && gz[l], ROVIT_ERR_NULL, "kan_stack_bwd: null pointer in layer %d", l);
    ROVIT_CHECK_ARG(dims[l] > 0 && dims[l + 1] > 0 && dims[l + 1] <= KB_MAXW, ROVIT_ERR_SHAPE, "kan_stack_bwd: layer %d is %d -> %d; widths after the input must be <= %d",
                    l, dims[l], dims[l + 1], KB_MAXW);
    ROVIT_CHECK_ARG(n_knots[l] >= 8 && n_knots[l] <= KAN_MAX_KNOTS, ROVIT_ERR_SHAPE, "kan_stack_bwd: bad knot count %d", n_knots[l]);
    a.W[l] = spline_w[l]; a.knots[l] = knots[l]; a.lw[l] = lin_w[l]; a.y[l] = outs[l]; a.gy[l] = grad_outs[l]; a.gz[l] = gz[l];
    a.nk[l] = n_knots[l]; a.act[l] = acts[l];
    if (want_dw) {
      ROVIT_CHECK_ARG(d_lin_w && d_lin_b && d_spline_w[l] && d_lin_w[l] && d_lin_b[l], ROVIT_ERR_NULL, "kan_stack_bwd: parameter gradients must be given together");
      a.dW[l] = d_spline_w[l]; a.dlw[l] = d_lin_w[l]; a.dlb[l] = d_lin_b[l];
      const int nb = n_knots[l] - 4;
      int bc = (24 * 1024) / (nb + 1 + dims[l + 1]);
      bc = bc > batch ? batch : bc;
      a.bc[l] = bc;
      a.thr[l] = dims[l + 1] * (nb + 2) >= 512 ? 1024 : 256;
      const size_t need = (KAN_MAX_KNOTS + (size_t)bc * (nb + 1 + dims[l + 1]) + 1024) * sizeof(float);
      lds_dw = need > lds_dw ? need : lds_dw;
    }
    a.wg0[l] = wgs;
    wgs += dims[l];
  }
  a.wg0[n_layers] = wgs;
  ROVIT_CHECK_ARG(grad_outs[n_layers - 1] != nullptr, ROVIT_ERR_NULL, "kan_stack_bwd: the gradient of the stack output is missing");
  // one sample per workgroup up to 1024 samples (every CU gets work at the benchmark's batch sizes), more beyond
  const int tb = batch <= 1024 ? 1 : (batch <= 4096 ? 4 : 16);
  const size_t lds_dx = (KAN_MAX_KNOTS + (size_t)3 * tb * KB_MAXW) * sizeof(float);
  hipLaunchKernelGGL(kan_stack_bwd_dx_kernel, dim3((batch + tb - 1) / tb), dim3(1024), lds_dx, (hipStream_t)stream, a, tb);
  ROVIT_CHECK_LAUNCH("kan_stack_bwd_dx_kernel");
  if (want_dw) {
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)kan_stack_bwd_dw_kernel, (size_t)(104 * 1024)), ROVIT_ERR_LAUNCH, "kan_stack_bwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(kan_stack_bwd_dw_kernel, dim3(wgs), dim3(1024), lds_dw, (hipStream_t)stream, a);
    ROVIT_CHECK_LAUNCH("kan_stack_bwd_dw_kernel");
  }
  return ROVIT_OK;
}

extern "C" int rovit_linear_fwd(const float* x, const float* w, const float* bias, const float* mask, float* y, int batch,
                                int in_f, int out_f, int flags, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && w && y, ROVIT_ERR_NULL, "linear_fwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && in_f > 0 && out_f > 0 && in_f % 4 == 0, ROVIT_ERR_SHAPE, "linear_fwd: bad shape (in %% 4)");
  ROVIT_CHECK_ARG(rovit_aligned16(x) && rovit_aligned16(w), ROVIT_ERR_ALIGN, "linear_fwd: x/w must be 16-byte aligned");
  const int n = batch * out_f;
  hipLaunchKernelGGL(lin_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, bias, mask, y, batch,
                     in_f, out_f, flags);
  ROVIT_CHECK_LAUNCH("lin_fwd_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_linear_bwd(const float* x, const float* w, const float* grad_y, const float* y_clamped,
                                const float* dx_mul, const float* dx_pos, float* dx, float* dw, float* db, int batch,
                                int in_f, int out_f, int accumulate_dx, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(x && w && grad_y, ROVIT_ERR_NULL, "linear_bwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && in_f > 0 && out_f > 0, ROVIT_ERR_SHAPE, "linear_bwd: bad shape");
  if (dx) {
    const int n = batch * in_f;
    hipLaunchKernelGGL(lin_bwd_dx_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_y, w, y_clamped,
                       dx_mul, dx_pos, dx, batch, in_f, out_f, accumulate_dx);
    ROVIT_CHECK_LAUNCH("lin_bwd_dx_kernel");
  }
  if (dw) {
    ROVIT_CHECK_ARG(db, ROVIT_ERR_NULL, "linear_bwd: dw and db must be given together");
    const int n = out_f * in_f;
    hipLaunchKernelGGL(lin_bwd_dw_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_y, x, y_clamped,
                       dw, db, batch, in_f, out_f);
    ROVIT_CHECK_LAUNCH("lin_bwd_dw_kernel");
  }
  return ROVIT_OK;
}

// ---- the three heads as one call (models/rovit_kan.py:93-116) --------------------------------------
// params (fp32, reference layouts): [0] cls.fc1.w [1] cls.fc1.b [2] cls.fc2.w [3] cls.fc2.b
//   [4] ord.fc1.w [5] ord.fc1.b [6] ord.fc2.w [7] ord.fc2.b
//   [8] unc.fc1.w [9] unc.fc1.b [10] unc.fc_mu.w [11] unc.fc_mu.b [12] unc.fc_logvar.w [13] unc.fc_logvar.b
// hidden: (3, B, hid) post-ReLU/dropout activations (saved for backward).  masks[h] may be NULL (eval).
extern "C" int rovit_heads_fwd(const float* features, const float* const* params, const float* const* masks, float* hidden,
                               float* cls_logits, float* ordinal_logits, float* mu, float* log_var, int batch, int embed, int hid,
                               int num_classes, int stage, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(features && params && hidden && cls_logits, ROVIT_ERR_NULL, "heads_fwd: null pointer");
  ROVIT_CHECK_ARG(stage >= 1 && stage <= 4, ROVIT_ERR_SHAPE, "heads_fwd: curriculum stage %d not in 1..4", stage);
  ROVIT_CHECK_ARG(batch > 0 && embed % 4 == 0 && hid % 4 == 0, ROVIT_ERR_SHAPE, "heads_fwd: bad shape");
  ROVIT_CHECK_ARG(stage < 2 || ordinal_logits, ROVIT_ERR_NULL, "heads_fwd: ordinal_logits is NULL at stage %d", stage);
  ROVIT_CHECK_ARG(stage < 3 || (mu && log_var), ROVIT_ERR_NULL, "heads_fwd: mu/log_var is NULL at stage %d", stage);
  const size_t hs = (size_t)batch * hid;
  const int nheads = stage >= 3 ? 3 : (stage >= 2 ? 2 : 1);
  LinFwdBatch f1{}; f1.B = batch;
  int work[4];
  for (int h = 0; h < nheads; ++h) {
    f1.d[h] = {features, params[4 * h], params[4 * h + 1], masks ? masks[h] : nullptr, hidden + h * hs, embed, hid, ROVIT_LIN_RELU};
    work[h] = batch * hid;
  }
  int rc = launch_lin_batch(f1, nheads, work, lin_fwd_batch_kernel, "lin_fwd_batch_kernel", (hipStream_t)stream, 64);
  if (rc) return rc;
  LinFwdBatch f2{}; f2.B = batch;
  int n = 0;
  f2.d[n] = {hidden, params[2], params[3], nullptr, cls_logits, hid, num_classes, 0}; work[n++] = batch * num_classes;
  if (stage >= 2) { f2.d[n] = {hidden + hs, params[6], params[7], nullptr, ordinal_logits, hid, num_classes - 1, 0}; work[n++] = batch * (num_classes - 1); }
  if (stage >= 3) {
    f2.d[n] = {hidden + 2 * hs, params[10], params[11], nullptr, mu, hid, 1, 0}; work[n++] = batch;
    f2.d[n] = {hidden + 2 * hs, params[12], params[13], nullptr, log_var, hid, 1, ROVIT_LIN_CLAMP10}; work[n++] = batch;
  }
  return launch_lin_batch(f2, n, work, lin_fwd_batch_kernel, "lin_fwd_batch_kernel", (hipStream_t)stream, 64);
}

// grads[] mirrors params[]; g_* may be NULL (head inactive or output unused); log_var is the clamped forward output.
// scratch: (3, B, hid) floats.  d_features is overwritten (accumulate_dfeat == 0) or accumulated into.
extern "C" int rovit_heads_bwd(const float* features, const float* const* params, const float* const* masks,
                               const float* hidden, const float* log_var, const float* g_cls, const float* g_ord,
                               const float* g_mu, const float* g_lv, float* d_features, float* const* grads, float* scratch,
                               int batch, int embed, int hid, int num_classes, int accumulate_dfeat, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(features && params && hidden && grads && scratch && d_features, ROVIT_ERR_NULL, "heads_bwd: null pointer");
  ROVIT_CHECK_ARG((g_mu == nullptr) == (g_lv == nullptr), ROVIT_ERR_NULL, "heads_bwd: mu and log_var gradients come together");
  ROVIT_CHECK_ARG(!g_lv || log_var, ROVIT_ERR_NULL, "heads_bwd: log_var output needed for the clamp gate");
  const size_t hs = (size_t)batch * hid;
  hipStream_t st = (hipStream_t)stream;
  const bool act[3] = {g_cls != nullptr, g_ord != nullptr, g_mu != nullptr};
  // 1) dL/d(hidden) of every active head (through ReLU and the dropout mask)
  LinDxBatch dxh{}; dxh.B = batch;
  LinDwBatch dwo{}; dwo.B = batch;
  int wx[4], ww[4], nx = 0, nw = 0;
  if (act[0]) {
    dxh.d[nx] = {1, {g_cls}, {params[2]}, {nullptr}, {num_classes}, masks ? masks[0] : nullptr, hidden, scratch, hid, 0}; wx[nx++] = batch * hid;
    dwo.d[nw] = {g_cls, hidden, nullptr, grads[2], grads[3], hid, num_classes}; ww[nw++] = num_classes * hid;
  }
  if (act[1]) {
    dxh.d[nx] = {1, {g_ord}, {params[6]}, {nullptr}, {num_classes - 1}, masks ? masks[1] : nullptr, hidden + hs, scratch + hs, hid, 0}; wx[nx++] = batch * hid;
    dwo.d[nw] = {g_ord, hidden + hs, nullptr, grads[6], grads[7], hid, num_classes - 1}; ww[nw++] = (num_classes - 1) * hid;
  }
  if (act[2]) {
    dxh.d[nx] = {2, {g_mu, g_lv}, {params[10], params[12]}, {nullptr, log_var}, {1, 1}, masks ? masks[2] : nullptr, hidden + 2 * hs,
                 scratch + 2 * hs, hid, 0};
    wx[nx++] = batch * hid;
    dwo.d[nw] = {g_mu, hidden + 2 * hs, nullptr, grads[10], grads[11], hid, 1}; ww[nw++] = hid;
    dwo.d[nw] = {g_lv, hidden + 2 * hs, log_var, grads[12], grads[13], hid, 1}; ww[nw++] = hid;
  }
  if (nx == 0) {
    if (!accumulate_dfeat) {
      hipError_t e = hipMemsetAsync(d_features, 0, (size_t)batch * embed * sizeof(float), st);
      ROVIT_CHECK_ARG(e == hipSuccess, ROVIT_ERR_LAUNCH, "heads_bwd: memset failed");
    }
    return ROVIT_OK;
  }
  int rc = launch_lin_batch(dxh, nx, wx, lin_bwd_dx_batch_kernel, "lin_bwd_dx_batch_kernel", st);
  if (rc) return rc;
  rc = launch_lin_batch(dwo, nw, ww, lin_bwd_dw_batch_kernel, "lin_bwd_dw_batch_kernel", st, 64, 1024);
  if (rc) return rc;
  // 2) through the first layers: d_features = sum_h dh_h W1_h ; dW1_h = dh_h^T features
  LinDxBatch dxf{}; dxf.B = batch;
  LinDwBatch dw1{}; dw1.B = batch;
  LinDxDesc& f = dxf.d[0];
  f = LinDxDesc{};
  f.mul = nullptr; f.pos = nullptr; f.dx = d_features; f.in_f = embed; f.accumulate = accumulate_dfeat;
  int n1 = 0, w1[4];
  for (int h = 0; h < 3; ++h) {
    if (!act[h]) continue;
    f.g[f.nsrc] = scratch + h * hs; f.w[f.nsrc] = params[4 * h]; f.yc[f.nsrc] = nullptr; f.out_f[f.nsrc] = hid; f.nsrc++;
    dw1.d[n1] = {scratch + h * hs, features, nullptr, grads[4 * h], grads[4 * h + 1], embed, hid}; w1[n1++] = hid * embed;
  }
  int wf[1] = {batch * embed};
  rc = launch_lin_batch(dxf, 1, wf, lin_bwd_dx_batch_kernel, "lin_bwd_dx_batch_kernel", st);
  if (rc) return rc;
  return launch_lin_batch(dw1, n1, w1, lin_bwd_dw_batch_kernel, "lin_bwd_dw_batch_kernel", st, 64, 1024);
}

// Parameter gradients of the whole head phase (head_phase.hip) in one launch.  Not part of the C ABI: called by
// rovit_head_phase_bwd behind its per-sample kernel, which has written dpre (heads) and gz (KAN layers).
int rovit_head_phase_dw_launch(const rovit_head_phase* p, hipStream_t st) {
  const int B = p->batch, E = p->embed, hid = p->hid, C = p->num_classes;
  const size_t hs = (size_t)B * hid;
  KanStackBwdArgs a{};
  int kan_wgs = 0;
  size_t lds = 0;
  if (p->kan_layers > 0 && p->g_kan) {
    a.x = p->features; a.B = B; a.nl = p->kan_layers; a.dx = nullptr;
    for (int l = 0; l <= p->kan_layers; ++l) a.dims[l] = p->kan_dims[l];
    for (int l = 0; l < p->kan_layers; ++l) {
      ROVIT_CHECK_ARG(p->kan_dw[l] && p->kan_dlw[l] && p->kan_dlb[l] && p->kan_gz[l], ROVIT_ERR_NULL,
                      "head_phase_bwd: parameter-gradient buffers of KAN layer %d missing", l);
      a.W[l] = p->kan_w[l]; a.knots[l] = p->kan_knots_p[l]; a.lw[l] = p->kan_lw[l]; a.y[l] = p->kan_out[l]; a.gy[l] = nullptr; a.gz[l] = p->kan_gz[l];
      a.nk[l] = p->kan_knots[l]; a.act[l] = p->kan_acts[l];
      a.dW[l] = p->kan_dw[l]; a.dlw[l] = p->kan_dlw[l]; a.dlb[l] = p->kan_dlb[l];
      const int nb = p->kan_knots[l] - 4;
      // batch rows per LDS chunk: ~32 KB per workgroup here (rovit_kan_stack_bwd takes 96 KB), because the head linears' workgroups of
      // the same grid reserve the same dynamic LDS and two 1024-thread workgroups should share a CU
      int bc = (7 * 1024) / (nb + 1 + p->kan_dims[l + 1]);
      bc = bc < 16 ? 16 : bc;
      bc = bc > B ? B : bc;
      a.bc[l] = bc;
      a.thr[l] = p->kan_dims[l + 1] * (nb + 2) >= 512 ? 1024 : 256;
      const size_t need = (KAN_MAX_KNOTS + (size_t)bc * (nb + 1 + p->kan_dims[l + 1]) + 1024) * sizeof(float);
      lds = need > lds ? need : lds;
      a.wg0[l] = kan_wgs;
      kan_wgs += p->kan_dims[l];
    }
    a.wg0[p->kan_layers] = kan_wgs;
  }
  LinDwBatch pb{};
  pb.B = B;
  int n = 0, blocks = 0;
  auto add = [&](const float* g, const float* x, const float* yc, float* dw, float* db, int in_f, int out_f) -> bool {
    if (!dw || !db) return false;
    pb.d[n] = {g, x, yc, dw, db, in_f, out_f};
    pb.first[n++] = blocks;
    blocks += (in_f * out_f + 63) / 64;
    return true;
  };
  const float* g_out[3] = {p->g_cls, p->g_ord, p->g_mu};
  const int nheads = p->stage >= 3 ? 3 : (p->stage >= 2 ? 2 : 1);
  for (int h = 0; h < nheads; ++h) {
    if (!g_out[h]) continue;
    ROVIT_CHECK_ARG(add(p->dpre + h * hs, p->features, nullptr, p->head_grads[4 * h], p->head_grads[4 * h + 1], E, hid), ROVIT_ERR_NULL,
                    "head_phase_bwd: fc1 gradient buffers of head %d missing", h);
    bool ok;
    if (h == 0) ok = add(p->g_cls, p->hidden, nullptr, p->head_grads[2], p->head_grads[3], hid, C);
    else if (h == 1) ok = add(p->g_ord, p->hidden + hs, nullptr, p->head_grads[6], p->head_grads[7], hid, C - 1);
    else ok = add(p->g_mu, p->hidden + 2 * hs, nullptr, p->head_grads[10], p->head_grads[11], hid, 1) &&
              add(p->g_lv, p->hidden + 2 * hs, p->lv, p->head_grads[12], p->head_grads[13], hid, 1);
    ROVIT_CHECK_ARG(ok, ROVIT_ERR_NULL, "head_phase_bwd: output-linear gradient buffers of head %d missing", h);
  }
  pb.first[n] = blocks;
  pb.n = n;
  if (kan_wgs + blocks == 0) return ROVIT_OK;
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)head_phase_dw_kernel, (size_t)(104 * 1024)), ROVIT_ERR_LAUNCH,
                  "head_phase_bwd: cannot raise the LDS limit");
  ROVIT_CHECK_ARG(lds <= 104 * 1024, ROVIT_ERR_SHAPE, "head_phase_bwd: the KAN weight-gradient tile needs %zu bytes of LDS", lds);
  hipLaunchKernelGGL(head_phase_dw_kernel, dim3(kan_wgs + blocks), dim3(1024), lds, st, a, pb, kan_wgs);
  ROVIT_CHECK_LAUNCH("head_phase_dw_kernel");
  return ROVIT_OK;
}
