// The head phase of a RoViT-KAN step in three launches (round 4).
//
// Reference being restated: what RoViTKAN.forward does with the backbone features (/root/reference/models/rovit_kan.py:88-124) --
//   ClassificationHead / OrdinalHead / UncertaintyHead.forward   models/heads.py:17-22, 38-43, 91-102   (curriculum gate rovit_kan.py:93-116)
//   KANSeverityModule.forward                                     models/kan.py:138-149 (KANLayer.forward :70-95, basis :8-44)
// -- and its autograd backward as reached from training/trainer.py:119,136.
//
// Before: 2 (dropout mask) + 2 (head linears) + 3 (KAN layers) launches forward, 2 + 4 + 1 backward, each a few microseconds of
// dependent latency on 256 samples: 0.06 + 0.11 ms of a 4.3 ms step spent in launches that leave the chip empty.  Here ONE workgroup
// owns ONE sample for the whole phase:
//   forward   head_phase_fwd_kernel     features row -> LDS; layer-0 basis; the 192 x 64 spline contraction (thread = (output, feature
//                                       slice), whole 32-byte basis rows of W[i, o, :] as two 16-byte loads against a dense basis
//                                       row in LDS: no prepared / transposed weight copy is needed) next to ONE dense product over
//                                       the rows [fc1 of every active head | KAN layer-0 linear] (4 lanes per row, 16-byte loads);
//                                       then the small layers (64 -> 16 -> 1) and the heads' output linears from LDS.
//   backward  head_phase_bwd_dx_kernel  the per-sample chain: heads' dpre, KAN gz of every layer top-down, the layer-0 spline term
//                                       and ONE transposed dense product over the same row list -> d_features (the sum of both
//                                       branches: no separate add launch).
//             head_phase_dw_kernel      (kan_heads.hip) every parameter gradient: the sample sums of the KAN stack and of the seven
//                                       head linears in one grid.
// Per workgroup the kernels read the ~0.74 MB of parameters from L2 once (a CU's L2 port, ~5 us, is the floor at one sample per
// CU); arithmetic is fp32 VALU in the order documented at each sum.
#include "common.h"
#include "kan_device.h"

namespace {

constexpr int HP_NT = 1024;
constexpr int HP_MAXW = 64;           // widest KAN layer behind the input
constexpr int HP_MAX_EMBED = 768, HP_MAX_HID = 256, HP_MAX_CLS = 8;

// Philox4x32-10 (Salmon et al. 2011): counter (c0, 0, offset lo, offset hi), key = seed
struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(unsigned long long seed, unsigned c0, unsigned long long offset) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  U4 c = {c0, 0u, (unsigned)offset, (unsigned)(offset >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}

// basis row of one input in LDS, 8 floats.  NBC = 7 / 8 (num_basis known at compile time; 7 is the reference's default, num_knots 5):
// the DENSE row d[k], so that a W[i, o, :] row is a plain dot product of whole-row loads (two instructions that do not depend on the
// interval index; the 28-byte rows of num_basis 7 are only dword-aligned: dword-aligned dwordx4 / dwordx3 loads, which gfx950 under
// ROCm executes in unaligned-access mode and hipcc emits for align-4 vector types).  NBC = 0 (any other num_basis): slots 0..3 the
// four non-zero values in the order of the four CONSECUTIVE weights they meet, slot 4 the first weight's index: one dword-aligned
// 16-byte load per (input, output) pair (four separate gathers of one dword each took the forward from 40 to 82 us at num_knots 32).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
template <int NBC>
__device__ __forceinline__ void hp_store_basis(float* dst, int j, const float* v) {
  if (NBC) {
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = k == j ? v[0] : (k == j - 1 ? v[1] : (k == j - 2 ? v[2] : (k == j - 3 ? v[3] : 0.f)));
    *(float4*)dst = make_float4(d[0], d[1], d[2], d[3]);
    *(float4*)(dst + 4) = make_float4(d[4], d[5], d[6], d[7]);
  } else {
    // the four live values ALIGNED to the four consecutive weights w[j0 .. j0 + 3], j0 = max(j, 3) - 3: u[k] pairs with w[j0 + k]
    // (j >= 3: u = v[3], v[2], v[1], v[0]; at the left edge, j < 3, the window starts at 0 and the missing terms are zeros)
    const int j0 = (j > 3 ? j : 3) - 3, sft = j - j0;          // sft = 3, or j at the left edge (-1: no live term, v is all zero)
    float u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = sft - k;
      u[k] = m == 0 ? v[0] : (m == 1 ? v[1] : (m == 2 ? v[2] : (m == 3 ? v[3] : 0.f)));
    }
    *(float4*)dst = make_float4(u[0], u[1], u[2], u[3]);
    dst[4] = __int_as_float(j0);
  }
}
// sum_k basis[k] W[k] for the (nb,) row at w
template <int NBC>
__device__ __forceinline__ float hp_dot_basis(const float* bas, const float* __restrict__ w) {
  if (NBC == 8) {
    const float4 w0 = *(const float4*)w, w1 = *(const float4*)(w + 4);
    const float4 d0 = *(const float4*)bas, d1 = *(const float4*)(bas + 4);
    float t = d0.x * w0.x;
    t = fmaf(d0.y, w0.y, t); t = fmaf(d0.z, w0.z, t); t = fmaf(d0.w, w0.w, t);
    t = fmaf(d1.x, w1.x, t); t = fmaf(d1.y, w1.y, t); t = fmaf(d1.z, w1.z, t); t = fmaf(d1.w, w1.w, t);
    return t;
  } else if (NBC == 7) {
    const f32x4u w0 = *(const f32x4u*)w;
    const f32x3u w1 = *(const f32x3u*)(w + 4);
    const float4 d0 = *(const float4*)bas, d1 = *(const float4*)(bas + 4);
    float t = d0.x * w0.x;
    t = fmaf(d0.y, w0.y, t); t = fmaf(d0.z, w0.z, t); t = fmaf(d0.w, w0.w, t);
    t = fmaf(d1.x, w1.x, t); t = fmaf(d1.y, w1.y, t); t = fmaf(d1.z, w1.z, t);
    return t;
  } else {
    const float4 u = *(const float4*)bas;
    const f32x4u wv = *(const f32x4u*)(w + __float_as_int(bas[4]));      // ONE dword-aligned 16-byte load of the four live weights
    float t = u.x * wv.x;
    t = fmaf(u.y, wv.y, t); t = fmaf(u.z, wv.z, t); t = fmaf(u.w, wv.w, t);
    return t;
  }
}

__device__ __forceinline__ int hp_nheads(int stage) { return stage >= 3 ? 3 : (stage >= 2 ? 2 : 1); }

// row r of the dense product: fc1 rows of the active heads, then the KAN layer-0 linear rows
__device__ __forceinline__ const float* hp_dense_row(const rovit_head_phase& p, int r, int hid, int nheads, int E) {
  const int h = r / hid;
  const float* base = h == 0 ? p.head_params[0] : (h == 1 ? p.head_params[4] : p.head_params[8]);
  if (h >= nheads) { base = p.kan_lw[0]; r -= nheads * hid; } else r -= h * hid;
  return base + (size_t)r * E;
}

struct HpLds {
  float *x, *bas, *f, *part, *h, *lin, *a0, *a1, *t, *knots, *dxs;
};
__device__ __forceinline__ HpLds hp_carve(float* smem, int E, int hid) {
  HpLds s;
  const int EB = E > HP_MAXW ? E : HP_MAXW;
  s.x = smem;                       // [E]           features row
  s.bas = s.x + E;                  // [8 EB]        basis rows of the current layer's inputs
  s.f = s.bas + 8 * EB;             // [EB]          backward: 1 - tanh^2
  s.part = s.f + EB;                // [HP_NT]       forward: spline partial sums [slice][out0]
  s.h = s.part + HP_NT;             // [3 hid]       hidden activations (forward) / dpre (backward)
  s.lin = s.h + 3 * hid;            // [64]          forward: layer-0 linear term; backward: gz of the current layer
  s.a0 = s.lin + HP_MAXW;           // [64]          ping-pong: activations (forward) / inter-layer gradients (backward)
  s.a1 = s.a0 + HP_MAXW;            // [64]
  s.t = s.a1 + HP_MAXW;             // [64 x 64]     terms of the small layers; backward: partial sums of the transposed dense product
  s.knots = s.t + HP_MAXW * HP_MAXW;   // [64]
  s.dxs = s.knots + KAN_MAX_KNOTS;  // [E]           backward: layer-0 spline term of d_features
  return s;
}
size_t hp_lds_bytes(int E, int hid) {
  const int EB = E > HP_MAXW ? E : HP_MAXW;
  return (size_t)(E + 8 * EB + EB + HP_NT + 3 * hid + 3 * HP_MAXW + HP_MAXW * HP_MAXW + KAN_MAX_KNOTS + E) * sizeof(float);
}

// ---- forward: a KAN layer behind the first (in, out <= 64): terms [in][out] in LDS, then one thread per output ----
template <int L, int NBC>
__device__ __forceinline__ void hp_fwd_small(const rovit_head_phase& p, const HpLds& s, const float* s_in, float* s_out, int b, int tid) {
  const int in = p.kan_dims[L], out = p.kan_dims[L + 1], nk = p.kan_knots[L], nb = nk - 4;
  __syncthreads();                                   // s_in complete; the previous layer is done with knots / bas / t
  if (tid < nk) s.knots[tid] = p.kan_knots_p[L][tid];
  __syncthreads();
  if (tid < in) {
    const Basis4 bs = kan_basis<false>(tanhf(s_in[tid]), s.knots, nk, 1.f / (s.knots[1] - s.knots[0]), nullptr);
    hp_store_basis<NBC>(s.bas + 8 * tid, bs.j, bs.v);
  }
  __syncthreads();
  const float* W = p.kan_w[L];
  const float* lw = p.kan_lw[L];
  for (int item = tid; item < in * out; item += HP_NT) {
    const int i = item / out, o = item - i * out;
    s.t[item] = fmaf(s_in[i], lw[(size_t)o * in + i], hp_dot_basis<NBC>(s.bas + 8 * i, W + ((size_t)i * out + o) * nb));
  }
  __syncthreads();
  if (tid < out) {
    float z = p.kan_lb[L][tid];
    for (int i = 0; i < in; ++i) z += s.t[i * out + tid];          // features in order
    const float a = act_apply(z, p.kan_acts[L]);
    s_out[tid] = a;
    p.kan_out[L][(size_t)b * out + tid] = a;
  }
}

template <int NBC>
__global__ __launch_bounds__(HP_NT) void head_phase_fwd_kernel(const rovit_head_phase p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int E = p.embed, hid = p.hid, C = p.num_classes, B = p.batch;
  const HpLds s = hp_carve(smem, E, hid);
  const bool kan = p.kan_layers > 0;
  const int nheads = hp_nheads(p.stage);
  const int out0 = kan ? p.kan_dims[1] : 0;

  if (kan && tid < p.kan_knots[0]) s.knots[tid] = p.kan_knots_p[0][tid];
  for (int i = tid; i < E; i += HP_NT) s.x[i] = p.features[(size_t)b * E + i];
  __syncthreads();
  if (kan) {
    const int nk = p.kan_knots[0];
    const float inv_h0 = 1.f / (s.knots[1] - s.knots[0]);
    for (int i = tid; i < E; i += HP_NT) {
      const Basis4 bs = kan_basis<false>(tanhf(s.x[i]), s.knots, nk, inv_h0, nullptr);
      hp_store_basis<NBC>(s.bas + 8 * i, bs.j, bs.v);
    }
  }
  __syncthreads();

  // (B) spline term of KAN layer 0: thread = (output o, feature slice); partial sums meet in LDS
  if (kan) {
    // Every workgroup reads the SAME weights: walking them in the same order at the same time would send all CUs of an XCD to one L2
    // channel at a time.  Workgroup b starts `b` slices (rows, below) further on; the partial sums keep their slice index, so the order
    // of the final sums -- and the result -- does not depend on the rotation.
    const int S = HP_NT / out0, fps = (E + S - 1) / S;
    const int sl0 = tid / out0, o = tid - sl0 * out0;
    if (sl0 < S) {
      const int sl = (sl0 + b) % S;
      const int nb = p.kan_knots[0] - 4;
      const int i0 = sl * fps, i1 = min(E, i0 + fps);
      const float* W = p.kan_w[0] + (size_t)o * nb;
      float acc = 0.f;
#pragma unroll 6
      for (int i = i0; i < i1; ++i) acc += hp_dot_basis<NBC>(s.bas + 8 * i, W + (size_t)i * out0 * nb);
      s.part[sl * out0 + o] = acc;
    }
  }
  // (C) dense rows [fc1 of the active heads | layer-0 linear]: four lanes per row, 16-byte loads, two cross-lane adds
  {
    const int R = nheads * hid + out0, E4 = E / 4;
    const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    const int rot = (b * 29) % R;
    for (int item = tid; item < R * 4; item += HP_NT) {
      const int part = item & 3;
      int r = (item >> 2) + rot;
      r = r >= R ? r - R : r;
      const float4* wrow = (const float4*)hp_dense_row(p, r, hid, nheads, E);
      float acc = 0.f;
#pragma unroll 6
      for (int q = part; q < E4; q += 4) {
        const float4 w = wrow[q], xv = ((const float4*)s.x)[q];
        acc = fmaf(w.x, xv.x, acc); acc = fmaf(w.y, xv.y, acc); acc = fmaf(w.z, xv.z, acc); acc = fmaf(w.w, xv.w, acc);
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) {
        const int h = r / hid;
        if (h < nheads) {
          const int k = r - h * hid;
          const float* b1 = h == 0 ? p.head_params[1] : (h == 1 ? p.head_params[5] : p.head_params[9]);
          const float* mk = h == 0 ? p.masks[0] : (h == 1 ? p.masks[1] : p.masks[2]);
          float v = fmaxf(acc + b1[k], 0.f);                                  // Linear -> ReLU (heads.py:18-19)
          if (mk) v *= mk[(size_t)b * hid + k];                               // -> Dropout (:20): given keep-mask ...
          else if (p.drop_p > 0.f) {                                          // ... or drawn here
            const U4 rr = philox4x32_10(p.seed, (unsigned)(b * hid + k), p.offset);
            const unsigned u = h == 0 ? rr.x : (h == 1 ? rr.y : rr.z);
            v = (float)(u >> 8) * (1.f / 16777216.f) < 1.f - p.drop_p ? v * inv_keep : 0.f;
          }
          s.h[r] = v;
          p.hidden[((size_t)h * B + b) * hid + k] = v;
        } else {
          s.lin[r - nheads * hid] = acc;
        }
      }
    }
  }
  __syncthreads();
  // (D) KAN layer-0 output; the heads' output linears (16 lanes per output row)
  if (kan && tid < out0) {
    const int S = HP_NT / out0;
    float z = p.kan_lb[0][tid] + s.lin[tid];
    for (int sl = 0; sl < S; ++sl) z += s.part[sl * out0 + tid];            // slices in order
    const float a = act_apply(z, p.kan_acts[0]);
    s.a0[tid] = a;
    p.kan_out[0][(size_t)b * out0 + tid] = a;
  }
  {
    const int R2 = C + (nheads >= 2 ? C - 1 : 0) + (nheads >= 3 ? 2 : 0), H4 = hid / 4;
    for (int item = tid; item < R2 * 16; item += HP_NT) {
      const int r = item >> 4, part = item & 15;
      int h = 0, j = r;
      if (r >= C) { h = 1; j = r - C; }
      if (nheads >= 2 && r >= 2 * C - 1) { h = 2; j = r - (2 * C - 1); }
      const float* w = h == 0 ? p.head_params[2] : (h == 1 ? p.head_params[6] : (j == 0 ? p.head_params[10] : p.head_params[12]));
      const float* bias = h == 0 ? p.head_params[3] : (h == 1 ? p.head_params[7] : (j == 0 ? p.head_params[11] : p.head_params[13]));
      const int jr = h == 2 ? 0 : j;
      const float4* wrow = (const float4*)(w + (size_t)jr * hid);
      const float4* hv = (const float4*)(s.h + h * hid);
      float acc = 0.f;
      for (int q = part; q < H4; q += 16) {
        const float4 a = wrow[q], c = hv[q];
        acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
      }
      acc = wave_sum16(acc);
      if (part == 0) {
        acc += bias[jr];
        if (h == 0) p.cls[(size_t)b * C + j] = acc;
        else if (h == 1) p.ord[(size_t)b * (C - 1) + j] = acc;
        else if (j == 0) p.mu[b] = acc;
        else p.lv[b] = fminf(fmaxf(acc, -10.f), 10.f);                       // heads.py:100
      }
    }
  }
  // (E) the small layers, activations ping-pong in LDS
  if (p.kan_layers > 1) hp_fwd_small<1, NBC>(p, s, s.a0, s.a1, b, tid);
  if (p.kan_layers > 2) hp_fwd_small<2, NBC>(p, s, s.a1, s.a0, b, tid);
  if (p.kan_layers > 3) hp_fwd_small<3, NBC>(p, s, s.a0, s.a1, b, tid);
}

// ---- backward: dL/dz of layer L from the gradient of its output, then (L >= 1) the gradient of its input ----
template <int L, int NBC>
__device__ __forceinline__ void hp_bwd_small(const rovit_head_phase& p, const HpLds& s, const float* s_gin, float* s_gout, int b, int tid) {
  const int in = p.kan_dims[L], out = p.kan_dims[L + 1], nk = p.kan_knots[L], nb = nk - 4;
  const bool top = L == p.kan_layers - 1;
  __syncthreads();                                   // s_gin complete; the layer above is done with knots / bas / t / lin
  if (tid < nk) s.knots[tid] = p.kan_knots_p[L][tid];
  if (tid < out) {
    const float g = top ? p.g_kan[(size_t)b * out + tid] : s_gin[tid];
    const float gz = act_grad(g, p.kan_out[L][(size_t)b * out + tid], p.kan_acts[L]);
    s.lin[tid] = gz;
    p.kan_gz[L][(size_t)b * out + tid] = gz;
  }
  __syncthreads();
  if (tid < in) {
    const float xn = tanhf(p.kan_out[L - 1][(size_t)b * in + tid]);
    float dv[4];
    const Basis4 bs = kan_basis<true>(xn, s.knots, nk, 1.f / (s.knots[1] - s.knots[0]), dv);
    hp_store_basis<NBC>(s.bas + 8 * tid, bs.j, dv);
    s.f[tid] = 1.f - xn * xn;                        // d tanh; the clamp is the identity on (-1, 1)
  }
  __syncthreads();
  const float* W = p.kan_w[L];
  const float* lw = p.kan_lw[L];
  for (int item = tid; item < in * out; item += HP_NT) {
    const int i = item / out, o = item - i * out;
    const float sp = hp_dot_basis<NBC>(s.bas + 8 * i, W + ((size_t)i * out + o) * nb);
    s.t[item] = s.lin[o] * fmaf(sp, s.f[i], lw[(size_t)o * in + i]);
  }
  __syncthreads();
  if (tid < in) {
    float g = 0.f;
    for (int o = 0; o < out; ++o) g += s.t[tid * out + o];              // outputs in order
    s_gout[tid] = g;
  }
}

template <int NBC>
__global__ __launch_bounds__(HP_NT) void head_phase_bwd_dx_kernel(const rovit_head_phase p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int E = p.embed, hid = p.hid, C = p.num_classes, B = p.batch;
  const HpLds s = hp_carve(smem, E, hid);
  const bool kan = p.kan_layers > 0 && p.g_kan;
  const int nheads = hp_nheads(p.stage);
  const int out0 = kan ? p.kan_dims[1] : 0;

  for (int i = tid; i < E; i += HP_NT) s.x[i] = p.features[(size_t)b * E + i];
  // heads: dL/d(hidden) through the output linears, then through dropout and ReLU (dpre is what the fc1 weight gradients read)
  {
    const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    for (int item = tid; item < nheads * hid; item += HP_NT) {
      const int h = item / hid, k = item - h * hid;
      float dh = 0.f;
      if (h == 0 && p.g_cls) {
        for (int j = 0; j < C; ++j) dh = fmaf(p.g_cls[(size_t)b * C + j], p.head_params[2][(size_t)j * hid + k], dh);
      } else if (h == 1 && p.g_ord) {
        for (int j = 0; j < C - 1; ++j) dh = fmaf(p.g_ord[(size_t)b * (C - 1) + j], p.head_params[6][(size_t)j * hid + k], dh);
      } else if (h == 2 && p.g_mu) {
        const float lv = p.lv[b];
        const float gl = (lv > -10.f && lv < 10.f) ? p.g_lv[b] : 0.f;          // clamp gate (heads.py:100)
        dh = fmaf(gl, p.head_params[12][k], p.g_mu[b] * p.head_params[10][k]);
      }
      const float* mk = h == 0 ? p.masks[0] : (h == 1 ? p.masks[1] : p.masks[2]);
      const float m = mk ? mk[(size_t)b * hid + k] : inv_keep;
      const float hv = p.hidden[((size_t)h * B + b) * hid + k];
      const float d = hv > 0.f ? dh * m : 0.f;
      s.h[item] = d;
      p.dpre[((size_t)h * B + b) * hid + k] = d;
    }
  }
  // KAN chain, top-down; the gradient between layers ping-pongs in LDS
  if (kan) {
    if (p.kan_layers > 3) hp_bwd_small<3, NBC>(p, s, s.a1, s.a0, b, tid);
    if (p.kan_layers > 2) hp_bwd_small<2, NBC>(p, s, s.a0, s.a1, b, tid);
    if (p.kan_layers > 1) hp_bwd_small<1, NBC>(p, s, s.a1, s.a0, b, tid);
    // layer 0: gz, then the derivative basis of the features
    const int nk = p.kan_knots[0];
    const bool top = p.kan_layers == 1;
    __syncthreads();
    if (tid < nk) s.knots[tid] = p.kan_knots_p[0][tid];
    if (tid < out0) {
      const float g = top ? p.g_kan[(size_t)b * out0 + tid] : s.a0[tid];
      const float gz = act_grad(g, p.kan_out[0][(size_t)b * out0 + tid], p.kan_acts[0]);
      s.lin[tid] = gz;
      p.kan_gz[0][(size_t)b * out0 + tid] = gz;
    }
    __syncthreads();
    if (p.d_features) {
      const float inv_h0 = 1.f / (s.knots[1] - s.knots[0]);
      for (int i = tid; i < E; i += HP_NT) {
        const float xn = tanhf(s.x[i]);
        float dv[4];
        const Basis4 bs = kan_basis<true>(xn, s.knots, nk, inv_h0, dv);
        hp_store_basis<NBC>(s.bas + 8 * i, bs.j, dv);
        s.f[i] = 1.f - xn * xn;
      }
    }
  }
  __syncthreads();
  if (!p.d_features) return;
  // layer-0 spline term: 16 lanes per feature, each a stride-16 set of outputs, whole basis rows of W[i, o, :]
  if (kan) {
    const int nb = p.kan_knots[0] - 4;
    const float* W = p.kan_w[0];
    const int roti = (b * 7) % E;
    for (int item = tid; item < E * 16; item += HP_NT) {
      const int part = item & 15;
      int i = (item >> 4) + roti;
      i = i >= E ? i - E : i;
      float acc = 0.f;
#pragma unroll 4
      for (int o = part; o < out0; o += 16) acc = fmaf(s.lin[o], hp_dot_basis<NBC>(s.bas + 8 * i, W + ((size_t)i * out0 + o) * nb), acc);
      acc = wave_sum16(acc);
      if (part == 0) s.dxs[i] = acc * s.f[i];
    }
  }
  // transposed dense product over the rows [fc1 of the heads | layer-0 linear]: thread = (16-byte column group, row slice)
  const int R = nheads * hid + out0, E4 = E / 4;
  const int S2 = HP_NT / E4, rps = (R + S2 - 1) / S2;
  {
    const int sl0 = tid / E4, c = tid - sl0 * E4;
    const int sl = (sl0 + b) % S2;
    if (sl0 < S2) {
      const int r0 = sl * rps, r1 = min(R, r0 + rps);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int r = r0; r < r1; ++r) {
        const float m = r < nheads * hid ? s.h[r] : s.lin[r - nheads * hid];
        const float4 w = ((const float4*)hp_dense_row(p, r, hid, nheads, E))[c];
        acc.x = fmaf(m, w.x, acc.x); acc.y = fmaf(m, w.y, acc.y); acc.z = fmaf(m, w.z, acc.z); acc.w = fmaf(m, w.w, acc.w);
      }
      *(float4*)(s.t + (size_t)sl * E + 4 * c) = acc;
    }
  }
  __syncthreads();
  for (int i = tid; i < E; i += HP_NT) {
    float g = kan ? s.dxs[i] : 0.f;
    for (int sl = 0; sl < S2; ++sl) g += s.t[sl * E + i];                  // row slices in order
    p.d_features[(size_t)b * E + i] = g;
  }
}

int hp_check(const rovit_head_phase* p, const char* who) {
  ROVIT_CHECK_ARG(p, ROVIT_ERR_NULL, "%s: null descriptor", who);
  ROVIT_CHECK_ARG(p->batch > 0 && p->embed >= 4 && p->embed <= HP_MAX_EMBED && p->embed % 4 == 0, ROVIT_ERR_SHAPE,
                  "%s: batch %d / embed %d (embed: multiple of 4, <= %d)", who, p->batch, p->embed, HP_MAX_EMBED);
  ROVIT_CHECK_ARG(p->hid >= 4 && p->hid <= HP_MAX_HID && p->hid % 4 == 0, ROVIT_ERR_SHAPE, "%s: hidden width %d (multiple of 4, <= %d)", who,
                  p->hid, HP_MAX_HID);
  ROVIT_CHECK_ARG(p->num_classes >= 2 && p->num_classes <= HP_MAX_CLS, ROVIT_ERR_SHAPE, "%s: %d classes (2..%d)", who, p->num_classes, HP_MAX_CLS);
  ROVIT_CHECK_ARG(p->stage >= 1 && p->stage <= 4, ROVIT_ERR_SHAPE, "%s: curriculum stage %d not in 1..4", who, p->stage);
  ROVIT_CHECK_ARG(p->kan_layers >= 0 && p->kan_layers <= 4, ROVIT_ERR_SHAPE, "%s: %d KAN layers (0..4)", who, p->kan_layers);
  ROVIT_CHECK_ARG(p->drop_p >= 0.f && p->drop_p < 1.f, ROVIT_ERR_SHAPE, "%s: dropout probability %g", who, (double)p->drop_p);
  ROVIT_CHECK_ARG(p->features && rovit_aligned16(p->features) && p->hidden, ROVIT_ERR_NULL, "%s: features (16-byte aligned) / hidden missing", who);
  const int nheads = p->stage >= 3 ? 3 : (p->stage >= 2 ? 2 : 1);
  for (int h = 0; h < nheads; ++h) {
    const int n = h == 2 ? 6 : 4;
    for (int q = 0; q < n; ++q)
      ROVIT_CHECK_ARG(p->head_params[4 * h + q] && rovit_aligned16(p->head_params[4 * h + q]), ROVIT_ERR_ALIGN,
                      "%s: head parameter %d missing or not 16-byte aligned", who, 4 * h + q);
  }
  if (p->kan_layers) {
    ROVIT_CHECK_ARG(p->kan_dims[0] == p->embed, ROVIT_ERR_SHAPE, "%s: the KAN stack's input width %d is not the feature width %d", who,
                    p->kan_dims[0], p->embed);
    for (int l = 0; l < p->kan_layers; ++l) {
      ROVIT_CHECK_ARG(p->kan_dims[l + 1] >= 1 && p->kan_dims[l + 1] <= HP_MAXW, ROVIT_ERR_SHAPE, "%s: KAN layer %d is %d wide (1..%d)", who, l,
                      p->kan_dims[l + 1], HP_MAXW);
      ROVIT_CHECK_ARG(p->kan_knots[l] >= 8 && p->kan_knots[l] <= KAN_MAX_KNOTS, ROVIT_ERR_SHAPE, "%s: KAN layer %d has %d knots (8..%d)", who, l,
                      p->kan_knots[l], KAN_MAX_KNOTS);
      ROVIT_CHECK_ARG(p->kan_w[l] && p->kan_knots_p[l] && p->kan_lw[l] && p->kan_lb[l] && p->kan_out[l], ROVIT_ERR_NULL,
                      "%s: KAN layer %d: null pointer", who, l);
      ROVIT_CHECK_ARG(rovit_aligned16(p->kan_w[l]) && rovit_aligned16(p->kan_lw[l]), ROVIT_ERR_ALIGN, "%s: KAN layer %d weights not 16-byte aligned", who, l);
    }
  }
  return ROVIT_OK;
}

// 7 / 8 when every layer has that num_basis (the dense-row kernels), else 0
int hp_nbc(const rovit_head_phase* p) {
  int nb = p->kan_layers ? p->kan_knots[0] - 4 : 7;
  for (int l = 1; l < p->kan_layers; ++l)
    if (p->kan_knots[l] - 4 != nb) nb = 0;
  return (nb == 7 || nb == 8) ? nb : 0;
}

#define HP_LAUNCH(kern, p, lds, stream)                                                                                          \
  do {                                                                                                                           \
    const int nbc__ = hp_nbc(p);                                                                                                 \
    if (nbc__ == 7) hipLaunchKernelGGL(kern<7>, dim3((p)->batch), dim3(HP_NT), lds, (hipStream_t)stream, *(p));                  \
    else if (nbc__ == 8) hipLaunchKernelGGL(kern<8>, dim3((p)->batch), dim3(HP_NT), lds, (hipStream_t)stream, *(p));             \
    else hipLaunchKernelGGL(kern<0>, dim3((p)->batch), dim3(HP_NT), lds, (hipStream_t)stream, *(p));                             \
  } while (0)

}  // namespace

int rovit_head_phase_dw_launch(const rovit_head_phase* p, hipStream_t st);      // kan_heads.hip

extern "C" int rovit_head_phase_fwd(const rovit_head_phase* p, rovit_stream_t stream) {
  const int rc = hp_check(p, "head_phase_fwd");
  if (rc) return rc;
  ROVIT_CHECK_ARG(p->cls && (p->stage < 2 || p->ord) && (p->stage < 3 || (p->mu && p->lv)), ROVIT_ERR_NULL,
                  "head_phase_fwd: head output missing at stage %d", p->stage);
  const size_t lds = hp_lds_bytes(p->embed, p->hid);
  HP_LAUNCH(head_phase_fwd_kernel, p, lds, stream);
  ROVIT_CHECK_LAUNCH("head_phase_fwd_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_head_phase_bwd(const rovit_head_phase* p, rovit_stream_t stream) {
  const int rc = hp_check(p, "head_phase_bwd");
  if (rc) return rc;
  ROVIT_CHECK_ARG(p->dpre, ROVIT_ERR_NULL, "head_phase_bwd: dpre scratch missing");
  ROVIT_CHECK_ARG((p->g_mu == nullptr) == (p->g_lv == nullptr), ROVIT_ERR_NULL, "head_phase_bwd: mu and log_var gradients come together");
  ROVIT_CHECK_ARG(!p->g_lv || p->lv, ROVIT_ERR_NULL, "head_phase_bwd: the log_var output is needed for the clamp gate");
  ROVIT_CHECK_ARG(!p->d_features || rovit_aligned16(p->d_features), ROVIT_ERR_ALIGN, "head_phase_bwd: d_features not 16-byte aligned");
  if (p->kan_layers && p->g_kan)
    for (int l = 0; l < p->kan_layers; ++l) ROVIT_CHECK_ARG(p->kan_gz[l], ROVIT_ERR_NULL, "head_phase_bwd: gz scratch of KAN layer %d missing", l);
  const size_t lds = hp_lds_bytes(p->embed, p->hid);
  HP_LAUNCH(head_phase_bwd_dx_kernel, p, lds, stream);
  ROVIT_CHECK_LAUNCH("head_phase_bwd_dx_kernel");
  if (p->want_param_grads) return rovit_head_phase_dw_launch(p, (hipStream_t)stream);
  return ROVIT_OK;
}

// The parameter-gradient launch alone (a caller that ran rovit_head_phase_bwd with want_param_grads == 0 on one stream and wants the
// sample sums on another, beside the backbone's backward: they are not on the path to d_features).
extern "C" int rovit_head_phase_bwd_params(const rovit_head_phase* p, rovit_stream_t stream) {
  const int rc = hp_check(p, "head_phase_bwd_params");
  if (rc) return rc;
  ROVIT_CHECK_ARG(p->dpre, ROVIT_ERR_NULL, "head_phase_bwd_params: dpre (written by rovit_head_phase_bwd) missing");
  return rovit_head_phase_dw_launch(p, (hipStream_t)stream);
}
