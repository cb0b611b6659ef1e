// Multi-head self-attention for DeiT-Tiny's 197-token sequences: softmax(Q K^T / 8) V, forward and backward.
//
// Reference arithmetic being restated: timm Attention.forward (qkv split, q*scale @ k^T, softmax, @ v)
// reached through /root/reference/models/backbone.py:12-25, and its autograd backward.
//
// One workgroup (7 waves) = one (image, head).  The whole 197x197 score tile stays on chip: K/V (and Q/dO in
// the backward) sit in LDS as 224 zero-padded rows; each wave owns 32 query rows (and 32 keys in backward).
// All matmuls are v_mfma_f32_16x16x32_bf16.  Scores are produced TRANSPOSED (keys on accumulator rows, the
// query on the lane) so that
//   * the softmax statistics of a query are lane-local (+ one xor-16/xor-32 exchange), and
//   * the bf16 probabilities are already the B operand of the P.V product: contraction slot (group g,
//     element j) of a 32-key step is key 16*(j>>2) + 4*g + (j&3), which is exactly where the accumulator
//     holds it; the matching V^T / K^T / Q^T / dO^T operands are gathered with ds_read_b64_tr_b16.
// No score/probability tile ever touches LDS or HBM.
#include "common.h"

namespace {

constexpr int HD = 64;               // head dim
constexpr int TP = 224;              // padded tokens (7 x 32)
constexpr int AST = HD + 16;         // LDS row stride in bf16: 160 bytes (odd multiple of 32 B)
constexpr int NW = 7;                // waves per workgroup
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
  const bf16* qkv;      // (B*T, 3*H*64): [q | k | v], head-major inside each third
  bf16* out;            // (B*T, H*64)
  float* lse2;          // (B, H, T): log2 sum_k exp(scale * q.k)
  int T, H;
  float scale;
  // backward
  const bf16* dout;     // (B*T, H*64)
  bf16* dqkv;           // (B*T, 3*H*64)
};

// stage rows [0, T) of a (T x 64) head slice into a [TP][AST] LDS tile, zero rows beyond T
__device__ __forceinline__ void stage_tile(bf16* dst, const bf16* src, int ld, int T, int tid) {
#pragma unroll
  for (int i = 0; i < TP * 8 / (NW * 64); ++i) {
    const int c = tid + i * NW * 64;
    const int row = c >> 3, kc = c & 7;
    const int rc = row < T ? row : T - 1;                       // clamped load + select: no branch around the load
    const bf16x8 v = keep_if(*(const bf16x8*)(src + (size_t)rc * ld + kc * 8), row < T);
    *(bf16x8*)(dst + row * AST + kc * 8) = v;
  }
}

// MFMA operand with rows/cols taken from LDS rows (natural contraction order over d)
__device__ __forceinline__ bf16x8 row_frag(const bf16* tile, int row, int ks, int lg) {
  return *(const bf16x8*)(tile + row * AST + ks * 32 + lg * 8);
}
// MFMA operand whose row/col index is the LDS COLUMN (d) and whose contraction runs over LDS rows
// r0 .. r0+31 in slot order 16*(j>>2) + 4*g + (j&3)
__device__ __forceinline__ bf16x8 col_frag(const bf16* tile, int r0, int dt, int l15, int lg) {
  const bf16* p = tile + (r0 + 4 * lg + (l15 >> 2)) * AST + dt * 16 + 4 * (l15 & 3);
  return cat4(lds_read_tr(p), lds_read_tr(p + 16 * AST));
}

// launch bound: 4 waves/SIMD (<= 128 VGPRs) so that TWO 7-wave workgroups share a CU (70 KB of LDS each) and one
// stages its K/V tiles while the other computes.
#ifndef ROVIT_LB_ATTN_FWD
#define ROVIT_LB_ATTN_FWD 2      // LDS (73 KB) admits two workgroups per CU; asking for four capped the kernel at 64 VGPRs with spills (step 5.99 -> 5.94 ms)
#endif
__global__ __launch_bounds__(NW * 64, ROVIT_LB_ATTN_FWD) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* Ks = lds;
  bf16* Vs = lds + TP * AST;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ld = 3 * a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  stage_tile(Ks, base + a.H * HD, ld, a.T, tid);
  stage_tile(Vs, base + 2 * a.H * HD, ld, a.T, tid);

  // this wave's 32 queries as the B operand (cols = query)
  bf16x8 qf[2][2];
  int qrow[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qrow[qt] = 32 * w + 16 * qt + l15;
    const int qc = qrow[qt] < a.T ? qrow[qt] : a.T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + lg * 8);
  }
  __syncthreads();

  // S^T[key][q] for 13 key tiles (208 >= 197 keys)
  f32x4 st[13][2];
#pragma unroll
  for (int kt = 0; kt < 13; ++kt) {
    const bf16x8 k0 = row_frag(Ks, 16 * kt + l15, 0, lg), k1 = row_frag(Ks, 16 * kt + l15, 1, lg);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = mfma16(k0, qf[qt][0], c);
      st[kt][qt] = mfma16(k1, qf[qt][1], c);
    }
  }
  const float c2 = a.scale * LOG2E;
  float inv_l[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    // VALU budget matters here (7 waves share 4 SIMDs): raw max on the unscaled scores, the scale folded into
    // one FMA in front of a bare v_exp_f32, and the key mask only on tiles that actually straddle T.
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      if (16 * kt + 16 > a.T) {                      // wave-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * kt + 4 * lg + r >= a.T) st[kt][qt][r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, st[kt][qt][r]);
    }
    m = group4_max(m);
    const float mc = m * c2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][qt][r], c2, -mc));
        st[kt][qt][r] = p;
        l += p;
      }
    l = group4_sum(l);
    inv_l[qt] = 1.f / l;
    if (lg == 0 && qrow[qt] < a.T && a.lse2) a.lse2[((size_t)b * a.H + h) * a.T + qrow[qt]] = mc + log2f(l);
  }

  // O^T[d][q] = sum_key V[key][d] P[q][key]
  f32x4 o[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 7; ++kb) {
    bf16x8 pf[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) pf[qt] = pack8(st[2 * kb][qt], kb < 6 ? st[kb < 6 ? 2 * kb + 1 : 0][qt] : zero4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 vf = col_frag(Vs, 32 * kb, dt, l15, lg);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) o[dt][qt] = mfma16(vf, pf[qt], o[dt][qt]);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (qrow[qt] < a.T) {
      bf16* dst = a.out + ((size_t)b * a.T + qrow[qt]) * (a.H * HD) + h * HD + 4 * lg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 v = o[dt][qt];
        v[0] *= inv_l[qt]; v[1] *= inv_l[qt]; v[2] *= inv_l[qt]; v[3] *= inv_l[qt];
        *(bf16x4*)(dst + 16 * dt) = pack4(v);
      }
    }
  }
}

// Backward.  Pass 1: wave w owns keys [32w, 32w+32) -> dK, dV (loop over query blocks).
//            Pass 2: wave w owns queries [32w, 32w+32) -> dQ (loop over key blocks).
// Each pass recomputes the probabilities it needs from Q, K and lse2 in the orientation that makes them the
// next MFMA's operand without any data movement, so there is no cross-wave reduction and no LDS traffic
// other than operand reads.
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* Qs = lds;
  bf16* Ks = Qs + TP * AST;
  bf16* Vs = Ks + TP * AST;
  bf16* Gs = Vs + TP * AST;                     // dO
  float* s_lse = (float*)(Gs + TP * AST);       // [TP]
  float* s_del = s_lse + TP;                    // [TP]  rowsum(dO * O)
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ld = 3 * a.H * HD, ldo = a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  const bf16* gbase = a.dout + (size_t)b * a.T * ldo + h * HD;
  const bf16* obase = a.out + (size_t)b * a.T * ldo + h * HD;
  stage_tile(Qs, base, ld, a.T, tid);
  stage_tile(Ks, base + a.H * HD, ld, a.T, tid);
  stage_tile(Vs, base + 2 * a.H * HD, ld, a.T, tid);
  stage_tile(Gs, gbase, ldo, a.T, tid);
  {
    const int row = tid >> 1, half = tid & 1;     // 448 threads = 224 rows x 2 halves
    float d = 0.f;
    if (row < a.T) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 g = *(const bf16x8*)(gbase + (size_t)row * ldo + half * 32 + i * 8);
        const bf16x8 o = *(const bf16x8*)(obase + (size_t)row * ldo + half * 32 + i * 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) d = fmaf((float)g[q], (float)o[q], d);
      }
    }
    d += __shfl_xor(d, 1);
    if (half == 0) {
      s_del[row] = d * a.scale;                 // pre-scaled: dS = P * (dP*scale - delta*scale)
      s_lse[row] = row < a.T ? a.lse2[((size_t)b * a.H + h) * a.T + row] : 0.f;
    }
  }
  __syncthreads();
  const float c2 = a.scale * LOG2E;

  // ---------------- pass 1: dK, dV for keys [32w, 32w+32) ----------------
  {
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[kt][ks] = row_frag(Ks, 32 * w + 16 * kt + l15, ks, lg);
        vf[kt][ks] = row_frag(Vs, 32 * w + 16 * kt + l15, ks, lg);
      }
    f32x4 dv[4][2], dk[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    bool kvalid[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) kvalid[kt] = (32 * w + 16 * kt + l15) < a.T;

    for (int qb = 0; qb < 7; ++qb) {
      f32x4 p[2][2], ds[2][2];     // [qt][kt]: rows q = 32qb + 16qt + 4lg + r, col key = 32w + 16kt + l15
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qr = 32 * qb + 16 * qt + l15;
        const bf16x8 q0 = row_frag(Qs, qr, 0, lg), q1 = row_frag(Qs, qr, 1, lg);
        const bf16x8 g0 = row_frag(Gs, qr, 0, lg), g1 = row_frag(Gs, qr, 1, lg);
        const float4 lse4 = *(const float4*)(s_lse + 32 * qb + 16 * qt + 4 * lg);
        const float4 del4 = *(const float4*)(s_del + 32 * qb + 16 * qt + 4 * lg);
        const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
        const float del_r[4] = {del4.x, del4.y, del4.z, del4.w};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(q0, kf[kt][0], s);  s = mfma16(q1, kf[kt][1], s);
          dp = mfma16(g0, vf[kt][0], dp); dp = mfma16(g1, vf[kt][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = kvalid[kt] ? __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_r[r])) : 0.f;
            p[qt][kt][r] = pr;
            ds[qt][kt][r] = pr * fmaf(dp[r], a.scale, -del_r[r]);
          }
        }
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { pf[kt] = pack8(p[0][kt], p[1][kt]); dsf[kt] = pack8(ds[0][kt], ds[1][kt]); }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 gT = col_frag(Gs, 32 * qb, dt, l15, lg);     // rows = d, slots = queries
        const bf16x8 qT = col_frag(Qs, 32 * qb, dt, l15, lg);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          dv[dt][kt] = mfma16(gT, pf[kt], dv[dt][kt]);            // dV^T[d][key]
          dk[dt][kt] = mfma16(qT, dsf[kt], dk[dt][kt]);           // dK^T[d][key]
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = 32 * w + 16 * kt + l15;
      if (key < a.T) {
        bf16* dst = a.dqkv + ((size_t)b * a.T + key) * ld + h * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          *(bf16x4*)(dst + a.H * HD + 16 * dt) = pack4(dk[dt][kt]);
          *(bf16x4*)(dst + 2 * a.H * HD + 16 * dt) = pack4(dv[dt][kt]);
        }
      }
    }
  }

  // ---------------- pass 2: dQ for queries [32w, 32w+32) ----------------
  {
    bf16x8 qf[2][2], gf[2][2];
    float lq[2], dq_[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      lq[qt] = s_lse[qr]; dq_[qt] = s_del[qr];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[qt][ks] = row_frag(Qs, qr, ks, lg); gf[qt][ks] = row_frag(Gs, qr, ks, lg); }
    }
    f32x4 dq[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < 7; ++kb) {
      f32x4 ds[2][2];             // [kt][qt]: rows key = 32kb + 16kt + 4lg + r, col q = 32w + 16qt + l15
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kr = 32 * kb + 16 * kt + l15;
        const bf16x8 k0 = row_frag(Ks, kr, 0, lg), k1 = row_frag(Ks, kr, 1, lg);
        const bf16x8 v0 = row_frag(Vs, kr, 0, lg), v1 = row_frag(Vs, kr, 1, lg);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(k0, qf[qt][0], s);  s = mfma16(k1, qf[qt][1], s);
          dp = mfma16(v0, gf[qt][0], dp); dp = mfma16(v1, gf[qt][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 32 * kb + 16 * kt + 4 * lg + r;
            const float pr = key < a.T ? __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lq[qt])) : 0.f;
            ds[kt][qt][r] = pr * fmaf(dp[r], a.scale, -dq_[qt]);
          }
        }
      }
      bf16x8 dsf[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dsf[qt] = pack8(ds[0][qt], ds[1][qt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kT = col_frag(Ks, 32 * kb, dt, l15, lg);     // rows = d, slots = keys
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = mfma16(kT, dsf[qt], dq[dt][qt]);   // dQ^T[d][q]
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      if (qr < a.T) {
        bf16* dst = a.dqkv + ((size_t)b * a.T + qr) * ld + h * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(dst + 16 * dt) = pack4(dq[dt][qt]);
      }
    }
  }
}

int check_attn(int batch, int tokens, int heads, int head_dim) {
  ROVIT_CHECK_ARG(batch > 0 && heads > 0, ROVIT_ERR_SHAPE, "attention: bad batch/heads");
  ROVIT_CHECK_ARG(head_dim == HD, ROVIT_ERR_SHAPE, "attention: head_dim must be %d (got %d)", HD, head_dim);
  ROVIT_CHECK_ARG(tokens > 0 && tokens <= 13 * 16, ROVIT_ERR_SHAPE, "attention: tokens must be <= 208 (got %d)", tokens);
  return ROVIT_OK;
}

// Explainability only (not on the training path): the softmax probabilities themselves, fp32 (B,H,T,T), from a saved
// qkv tensor.  One wave per query row; lanes stride over the keys.
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16* __restrict__ qkv, float* __restrict__ probs, int B, int T, int H,
                                                         int HD, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // (b, h, q)
  const int lane = threadIdx.x & 63;
  if (row >= B * H * T) return;
  const int q = row % T, h = (row / T) % H, b = row / (T * H);
  const int ld = 3 * H * HD;
  const bf16* qp = qkv + ((size_t)b * T + q) * ld + h * HD;
  float s[4];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int key = lane + 64 * u;
    s[u] = -INFINITY;
    if (key < T) {
      const bf16* kp = qkv + ((size_t)b * T + key) * ld + H * HD + h * HD;
      float acc = 0.f;
      for (int d = 0; d < HD; ++d) acc = fmaf((float)qp[d], (float)kp[d], acc);
      s[u] = acc * scale;
      mx = fmaxf(mx, s[u]);
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) { s[u] = lane + 64 * u < T ? __expf(s[u] - mx) : 0.f; sum += s[u]; }
  sum = wave_sum64(sum);
  const float inv = 1.f / sum;
  float* out = probs + (size_t)row * T;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (lane + 64 * u < T) out[lane + 64 * u] = s[u] * inv;
}

}  // namespace

extern "C" int rovit_attention_probs(const void* qkv, float* probs, int batch, int tokens, int heads, int head_dim, float scale,
                                     rovit_stream_t stream) {
  ROVIT_CHECK_ARG(qkv && probs, ROVIT_ERR_NULL, "attention_probs: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && tokens > 0 && tokens <= 256 && heads > 0 && head_dim > 0, ROVIT_ERR_SHAPE,
                  "attention_probs: unsupported shape (tokens <= 256)");
  const int rows = batch * heads * tokens;
  hipLaunchKernelGGL(attn_probs_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, probs, batch, tokens,
                     heads, head_dim, scale);
  ROVIT_CHECK_LAUNCH("attn_probs_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_attention_fwd(const void* qkv, void* out, float* lse2, int batch, int tokens, int heads, int head_dim,
                                   float scale, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(qkv && out, ROVIT_ERR_NULL, "attention_fwd: null pointer");
  int rc = check_attn(batch, tokens, heads, head_dim);
  if (rc) return rc;
  ROVIT_CHECK_ARG(rovit_aligned16(qkv) && rovit_aligned16(out), ROVIT_ERR_ALIGN, "attention_fwd: alignment");
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.lse2 = lse2; a.T = tokens; a.H = heads; a.scale = scale;
  const size_t lds = (size_t)2 * TP * AST * sizeof(bf16);
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_fwd_kernel, lds), ROVIT_ERR_LAUNCH, "attention_fwd: cannot raise the LDS limit");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(batch * heads), dim3(NW * 64), lds, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("attn_fwd_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_attention_bwd(const void* qkv, const void* out, const float* lse2, const void* dout, void* dqkv, int batch,
                                   int tokens, int heads, int head_dim, float scale, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(qkv && out && lse2 && dout && dqkv, ROVIT_ERR_NULL, "attention_bwd: null pointer");
  int rc = check_attn(batch, tokens, heads, head_dim);
  if (rc) return rc;
  ROVIT_CHECK_ARG(rovit_aligned16(qkv) && rovit_aligned16(out) && rovit_aligned16(dout) && rovit_aligned16(dqkv), ROVIT_ERR_ALIGN,
                  "attention_bwd: alignment");
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.lse2 = (float*)lse2; a.T = tokens; a.H = heads; a.scale = scale;
  a.dout = (const bf16*)dout; a.dqkv = (bf16*)dqkv;
  const size_t lds = (size_t)4 * TP * AST * sizeof(bf16) + 2 * TP * sizeof(float);
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_kernel, lds), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(batch * heads), dim3(NW * 64), lds, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("attn_bwd_kernel");
  return ROVIT_OK;
}
