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
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

#ifdef ROVIT_DEV
#define ATTN_DBG(a, bit) ((a).dbg & (bit))
#else
#define ATTN_DBG(a, bit) (false)    // the product kernels have no skip-work path
#endif

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
  int dout_rows;        // rows of every image's dout that carry gradient (T; 1 = only the class token's: the last block, whose other rows are treated as zero and never read)
  bf16* dqkv;           // (B*T, 3*H*64)
#ifdef ROVIT_DEV
  int dbg;              // developer library only (ROVIT_KNOB_ATTN_DBG, timing ablations): bit 0 skip pass 1, bit 1 skip pass 2
#endif
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

// Forward, round 4: TWO workgroups per CU.  The round-3 kernel kept the scores of BOTH 16-query tiles of a wave in registers
// (146 VGPRs -> 3 waves per SIMD -> 12 wave slots per CU: a second 7-wave workgroup never fitted, whatever the launch bound said,
// so every CU ran load -> compute -> store in lockstep with the rest of the chip).  Here a wave walks its two query tiles ONE AFTER
// THE OTHER (52 score registers instead of 104; <= 128 VGPRs under __launch_bounds__(448, 4)), so 14 waves = two (image, head)
// items share a CU (2 x 71.7 KB of LDS) and one stages its K / V tiles while the other computes.  The K / V fragments are read
// from LDS once per tile instead of once per wave (2 x 54 KB per wave: ~1.2 us of LDS time per item, hidden behind the MFMAs of
// the other waves); a wave whose second tile lies beyond T skips it (wave 6 at T = 197: 13 tiles of work instead of 14).
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_kernel(const AttnArgs a) {
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
  // the first tile's 16 queries as the B operand (cols = query); the second tile's are requested behind the S products
  bf16x8 qf[2];
  {
    const int qr = 32 * w + l15, qc = qr < a.T ? qr : a.T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + lg * 8);
  }
  __syncthreads();
  const float c2 = a.scale * LOG2E;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
    const int q0 = 32 * w + 16 * qt;
    if (q0 >= a.T) break;                             // wave-uniform: nothing of this tile is stored
    const int qrow = q0 + l15;
    // S^T[key][q] for 13 key tiles (208 >= 197 keys)
    f32x4 st[13];
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      const bf16x8 k0 = row_frag(Ks, 16 * kt + l15, 0, lg), k1 = row_frag(Ks, 16 * kt + l15, 1, lg);
      st[kt] = mfma16(k1, qf[1], mfma16(k0, qf[0], zero4));
    }
    if (qt == 0) {                                    // next tile's queries: in flight during the softmax and the P.V products
      const int qr = q0 + 16 + l15, qc = qr < a.T ? qr : a.T - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + lg * 8);
    }
    // VALU budget matters here: raw max on the unscaled scores, the scale folded into one FMA in front of a bare v_exp_f32, and
    // the key mask only on tiles that actually straddle T.
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      if (16 * kt + 16 > a.T) {                       // wave-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * kt + 4 * lg + r >= a.T) st[kt][r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, st[kt][r]);
    }
    m = group4_max(m);
    const float mc = m * c2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], c2, -mc));
        st[kt][r] = p;
        l += p;
      }
    l = group4_sum(l);
    const float inv_l = 1.f / l;
    if (lg == 0 && qrow < a.T && a.lse2) a.lse2[((size_t)b * a.H + h) * a.T + qrow] = mc + log2f(l);
    // O^T[d][q] = sum_key V[key][d] P[q][key]
    f32x4 o[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
      const bf16x8 pf = pack8(st[2 * kb], kb < 6 ? st[kb < 6 ? 2 * kb + 1 : 0] : zero4);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(col_frag(Vs, 32 * kb, dt, l15, lg), pf, o[dt]);
    }
    if (qrow < a.T) {
      bf16* dst = a.out + ((size_t)b * a.T + qrow) * (a.H * HD) + h * HD + 4 * lg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 v = o[dt];
        v[0] *= inv_l; v[1] *= inv_l; v[2] *= inv_l; v[3] *= inv_l;
        *(bf16x4*)(dst + 16 * dt) = pack4(v);
      }
    }
  }
}

#ifdef ROVIT_DEV   // measured no faster (below): developer library only, knob 20
// Forward, round 4 (late): THREE workgroups per CU = all 768 (image, head) items of a batch-256 launch resident at once (two per CU are 1.5
// rounds: a third of the launch runs with half the chip's slots empty).  What it takes:
//   * LDS: K and V as UNPADDED 208-row tiles (13 key tiles; 128-byte rows: 53 248 bytes per workgroup, 3 x = 159 744 of the CU's 163 840).
//     Unpadded rows alone put every second row on the same banks; the 16-byte chunk index is XOR-swizzled with f(row) = (bit 2 of the row) << 1
//     | (bit 1) << 2, found by brute force over the linear swizzles against the bank model of MI355X_MICROARCH.md (ds_read_b128: four groups of 16
//     lanes; ds_read_b64_tr_b16: two groups of 32): conflict-free for the row-fragment reads AND the transposed column-fragment reads -- the
//     same model calls the 160-byte padded rows conflict-free and unpadded unswizzled rows conflicted (tools/lds_swizzle_search.py);
//   * registers: <= 80 for six waves per SIMD.  The 52 score registers of a query tile go: the scores are computed TWICE -- a first sweep over
//     the 13 key tiles keeps only the running maximum, a second one recomputes each 32-key block's scores, exponentiates them against that
//     maximum and feeds the P V product at once.  +26 MFMAs and +26 fragment reads per query tile in a kernel whose matrix pipe is 14 % busy.
// Same products and maximum as attn_fwd_kernel; outputs within one bf16 ulp of it (tools/attn_fwd3_check.py).  MEASURED (tools/ab_attn_fwd3.sh, one box,
// batch 256): 27.6 / 26.8 us per launch against 27.3 / 26.2 for the two-workgroup kernel, whole steps 4.10 / 4.07 against 4.05 / 4.04 ms: NO
// gain -- the third workgroup's residency buys what the second score sweep costs.  (A first version that spilled 14-24 registers at the
// 80-register cap also produced wrong results; this one, 68 registers, does not spill.)
constexpr int TK3 = 208;                 // key rows in LDS (13 tiles of 16); T <= 208
#ifdef ATTN3_NOSWZ
__device__ __forceinline__ int swz3(int row) { return 0; }
#else
__device__ __forceinline__ int swz3(int row) { return (((row >> 2) & 1) << 1) | (((row >> 1) & 1) << 2); }
#endif
#ifndef ATTN3_LB
#define ATTN3_LB 6
#endif
__global__ __launch_bounds__(NW * 64, ATTN3_LB) void attn_fwd3_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* Ks = lds;
  bf16* Vs = lds + TK3 * HD;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ld = 3 * a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  // stage K and V: 208 rows x 8 chunks of 16 bytes each, rows beyond T zero
  for (int c = tid; c < TK3 * 8; c += NW * 64) {
    const int row = c >> 3, kc = c & 7;
    const int rc = row < a.T ? row : a.T - 1;
    const int dst = row * HD + ((kc ^ swz3(row)) << 3);
    *(bf16x8*)(Ks + dst) = keep_if(*(const bf16x8*)(base + a.H * HD + (size_t)rc * ld + kc * 8), row < a.T);
    *(bf16x8*)(Vs + dst) = keep_if(*(const bf16x8*)(base + 2 * a.H * HD + (size_t)rc * ld + kc * 8), row < a.T);
  }
  // Per-lane fragment offsets, computed once: the swizzle reads bits 1-2 of the row only, and neither 16 kt (row fragments) nor 32 kb
  // (column fragments) touches them.  Row fragment (key tile kt, k-step ks): Ks + 16 kt HD + koff[ks]; column fragment (key block kb, d-tile dt):
  // Vs + 32 kb HD + voff[dt] and the same + 16 HD.
  int koff[2], voff[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = l15 * HD + (((4 * ks + lg) ^ swz3(l15)) << 3);
  {
    const int rl = 4 * lg + (l15 >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) voff[dt] = rl * HD + (((2 * dt + ((l15 >> 1) & 1)) ^ swz3(rl)) << 3) + 4 * (l15 & 1);
  }
  __syncthreads();
  const float c2 = a.scale * LOG2E;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
    const int q0 = 32 * w + 16 * qt;
    if (q0 >= a.T) break;                             // wave-uniform: nothing of this tile is stored
    const int qrow = q0 + l15;
    bf16x8 qf[2];
    {
      const int qc = qrow < a.T ? qrow : a.T - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + lg * 8);
    }
    auto scores = [&](int kt) -> f32x4 {
      const bf16* kp = Ks + 16 * kt * HD;
      f32x4 s = mfma16(*(const bf16x8*)(kp + koff[1]), qf[1], mfma16(*(const bf16x8*)(kp + koff[0]), qf[0], zero4));
      if (16 * kt + 16 > a.T) {                       // wave-uniform: the tile straddles T
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * kt + 4 * lg + r >= a.T) s[r] = -INFINITY;
      }
      return s;
    };
    // sweep 1: the row maximum
    float m = -INFINITY;
#pragma unroll 1
    for (int kt = 0; kt < 13; ++kt) {
      const f32x4 s = scores(kt);
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, s[r]);
    }
    m = group4_max(m);
    const float mc = m * c2;
    // sweep 2: scores again, block by block; exponentials; P V
    float l = 0.f;
    f32x4 o[4] = {zero4, zero4, zero4, zero4};
    auto block = [&](int kb, auto last) {
      constexpr bool LAST = decltype(last)::value;       // key block 6: keys 192 .. 207 only (no tile 13, and no V rows beyond 207)
      f32x4 p0 = scores(2 * kb), p1 = zero4;
#pragma unroll
      for (int r = 0; r < 4; ++r) { p0[r] = __builtin_amdgcn_exp2f(fmaf(p0[r], c2, -mc)); l += p0[r]; }
      if (!LAST) {
        p1 = scores(2 * kb + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) { p1[r] = __builtin_amdgcn_exp2f(fmaf(p1[r], c2, -mc)); l += p1[r]; }
      }
      const bf16x8 pf = pack8(p0, p1);
      const bf16* vp = Vs + 32 * kb * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x4 lo = lds_read_tr(vp + voff[dt]);
        const bf16x4 hi = lds_read_tr(LAST ? vp + voff[dt] : vp + voff[dt] + 16 * HD);      // LAST: those keys carry p = 0; read rows that exist
        o[dt] = mfma16(cat4(lo, hi), pf, o[dt]);
      }
    };
#pragma unroll 1
    for (int kb = 0; kb < 6; ++kb) block(kb, std::false_type());
    block(6, std::true_type());
    l = group4_sum(l);
    const float inv_l = 1.f / l;
    if (lg == 0 && qrow < a.T && a.lse2) a.lse2[((size_t)b * a.H + h) * a.T + qrow] = mc + log2f(l);
    if (qrow < a.T) {
      bf16* dst = a.out + ((size_t)b * a.T + qrow) * (a.H * HD) + h * HD + 4 * lg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 v = o[dt];
        v[0] *= inv_l; v[1] *= inv_l; v[2] *= inv_l; v[3] *= inv_l;
        *(bf16x4*)(dst + 16 * dt) = pack4(v);
      }
    }
  }
}

#endif  // ROVIT_DEV (attn_fwd3_kernel)

#ifdef ROVIT_DEV   // round 3's forward (both query tiles of a wave in registers, one workgroup per CU): A/B in the developer build only
// launch bound: 4 waves/SIMD (<= 128 VGPRs) so that TWO 7-wave workgroups share a CU (70 KB of LDS each) and one
// stages its K/V tiles while the other computes.
#ifndef ROVIT_LB_ATTN_FWD
#define ROVIT_LB_ATTN_FWD 2      // LDS (73 KB) admits two workgroups per CU; asking for four capped the kernel at 64 VGPRs with spills (step 5.99 -> 5.94 ms)
#endif
__global__ __launch_bounds__(NW * 64, ROVIT_LB_ATTN_FWD) void attn_fwd_kernel_r3(const AttnArgs a) {
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

#endif  // ROVIT_DEV

// Transposed operand reads ISSUED EARLY (inline asm: hipcc otherwise sinks each ds_read_b64_tr_b16 pair to just in front of the MFMA
// that uses it, and the dV / dK step of pass 1 then runs at LDS latency: ~1 170 cycles per query block for 256 cycles of MFMAs).  The
// asm reads are invisible to the compiler's lgkmcnt bookkeeping (extra outstanding reads only make its own counted waits more
// conservative), so tr_wait() -- lgkmcnt(0) + a scheduling fence -- must stand between tr_issue() and the first use of tr_val().
struct TrFrag { bf16x4 lo, hi; };
// Column order: output tile dt of the product holds the tile's LDS columns 32 (dt >> 1) + 8 i + 4 (dt & 1) + j on its rows 4 i + j, so that a
// lane's accumulators of the tile pair (2k, 2k + 1) are EIGHT CONSECUTIVE head-dim values 32 k + 8 lg .. + 7 of its row: one 16-byte store
// instead of two 8-byte ones (the dK / dV / dQ stores cost ~6 us of the launch in 24 eight-byte store instructions per wave).
__device__ __forceinline__ TrFrag tr_issue(const bf16* tile, int r0, int dt, int l15, int lg) {
  const bf16* p = tile + (r0 + 4 * lg + (l15 >> 2)) * AST + 32 * (dt >> 1) + 8 * (l15 & 3) + 4 * (dt & 1);
  const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) bf16*)p;
  TrFrag f;
  static_assert(16 * AST * sizeof(bf16) == 2560, "offset of the second row block");
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2560" : "=&v"(f.lo), "=&v"(f.hi) : "v"(a0) : "memory");
  return f;
}
// the same operand, compiler-scheduled (the round-4 kernels hide the read latency with four waves per SIMD)
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int r0, int dt, int l15, int lg) {
  const bf16* p = tile + (r0 + 4 * lg + (l15 >> 2)) * AST + 32 * (dt >> 1) + 8 * (l15 & 3) + 4 * (dt & 1);
  return cat4(lds_read_tr(p), lds_read_tr(p + 16 * AST));
}
__device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 tr_val(const TrFrag& f) { return cat4(f.lo, f.hi); }

// Backward, round 4: TWO workgroups per CU and TWO workgroups per (image, head).  Round 3: 237 VGPRs and four LDS tiles = 143 KB, one
// workgroup per CU, so the whole chip loaded, computed and stored in lockstep (0.34 of HBM peak, HBM idle two thirds of the launch).
//   Pass 1 (its own workgroup): wave w owns keys [32w, 32w+32) -> dK, dV (loop over query blocks); needs the Q and dO tiles in LDS
//           and its own K / V rows as register fragments (read straight from global memory).
//   Pass 2 (its own workgroup): wave w owns queries [32w, 32w+32) -> dQ (loop over key blocks); needs the K and V tiles in LDS and
//           its own Q / dO rows.
// The passes share nothing but the row statistics (lse, delta = rowsum(dO O): recomputed by both, 50 KB of reads that hit L2), so a
// workgroup stages only TWO tiles (73.5 KB) and, with <= 128 VGPRs, FOUR waves per SIMD = two workgroups per CU: 2 x 768 workgroups
// in three even rounds of 512 (768 two-pass workgroups were 1.5 rounds: the last third ran alone on its CU), a pass-1 and a pass-2
// workgroup of different length side by side, so loads, matrix work and stores of the chip no longer move in phase.
//   * pass 1 walks the wave's two 16-key sub-tiles ONE AFTER THE OTHER (accumulators 32 registers instead of 64); price: the Q / dO
//     operand fragments are read from LDS once per sub-tile;
//   * pass 2 keeps both 16-query sub-tiles in registers (32 accumulators), K / V fragments read once per block;
//   * transposed operands in the conflict-free column order (16 dt + 4 p: round 3's pair order, which made a lane's two tiles eight
//     consecutive head-dim values, spread every ds_read_b64_tr_b16 over half-used bank rows: 2-way conflicts on every transposed
//     read, 23 % of the LDS cycles); the 16-byte stores come from one v_permlane16_swap per packed register pair instead;
//   * sub-tiles and half blocks that lie beyond T are skipped (13 of 14 sub-tiles and 13 of 14 half blocks at T = 197).
// The arithmetic per stored element is unchanged: results are bit-identical to the round-3 kernel (tools/ab_attention.py).
// Each pass recomputes the probabilities it needs from Q, K and lse2 in the orientation that makes them the next MFMA's operand
// without any data movement, so there is no cross-wave reduction and no LDS traffic other than operand reads.
__device__ __forceinline__ bf16x8 global_row_frag(const bf16* base, int ld, int row, int T, int ks, int lg) {
  const int rc = row < T ? row : T - 1;
  return keep_if(*(const bf16x8*)(base + (size_t)rc * ld + ks * 32 + lg * 8), row < T);
}
// Accumulator tiles t0 = tile 2k, t1 = tile 2k+1 of a 16-column output (lane group lg holds head-dim rows 16 dt + 4 lg + r of column
// l15): after one v_permlane16_swap per packed register (odd 16-lane rows of the first operand <-> even rows of the second) an even
// lane group holds tile 2k's values 8 (lg >> 1) .. + 7 and an odd one tile 2k+1's: ONE 16-byte store per lane.
__device__ __forceinline__ void store_tile_pair(bf16* row_base, f32x4 t0, f32x4 t1, int k, int lg) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const u32x2 x = __builtin_bit_cast(u32x2, pack4(t0)), y = __builtin_bit_cast(u32x2, pack4(t1));
  const auto s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
  u32x4 v;
  v[0] = s0[0]; v[1] = s1[0]; v[2] = s0[1]; v[3] = s1[1];
  *(u32x4*)(row_base + 16 * (2 * k + (lg & 1)) + 8 * (lg >> 1)) = v;
}
template <bool SPLIT>
__global__ __launch_bounds__(NW * 64, 4) void attn_bwd_kernel(const AttnArgs a, int n_items) {
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* T0 = lds;                                // pass 1: Q,  pass 2: K
  bf16* T1 = T0 + TP * AST;                      // pass 1: dO, pass 2: V
  float* s_lse = (float*)(T1 + TP * AST);        // [TP]
  float* s_del = s_lse + TP;                     // [TP]  -rowsum(dO * O)
  // Workgroup id -> (item, pass).  Ids b, b + 8, ... share an XCD (round-robin dispatch) and are dealt to its 32 CUs in turn, so
  // `pass = (id >> 3) & 1` put every pass-1 workgroup on an even CU and every (lighter) pass-2 workgroup on an odd one: 66 us.
  // Here the j-th workgroup of an XCD (j = id >> 3) belongs to group j >> 6 of 32 items; the first 32 of a group run pass 1, the
  // next 32 pass 2 of the same items: one of each per CU in the first round, and an item's two workgroups share the XCD's L2.
  // Speed only: any placement gives the same results.
  const int j = blockIdx.x >> 3;
  const int pass = SPLIT ? (j >> 5) & 1 : 0;     // !SPLIT: one workgroup per item runs pass 1, re-stages the same LDS with K / V, runs pass 2
  const int bh = SPLIT ? ((j >> 6) * 32 + (j & 31)) * 8 + (blockIdx.x & 7) : (int)blockIdx.x;
  if (bh >= n_items) return;                     // whole workgroup
  const int b = bh / a.H, h = bh - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: the sub-tile skips below become scalar branches
  const int l15 = lane & 15, lg = lane >> 4;
  const int ld = 3 * a.H * HD, ldo = a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  const bf16* gbase = a.dout + (size_t)b * a.T * ldo + h * HD;
  const bf16* obase = a.out + (size_t)b * a.T * ldo + h * HD;
  if (pass == 0) {
    stage_tile(T0, base, ld, a.T, tid);
    stage_tile(T1, gbase, ldo, a.dout_rows, tid);          // rows beyond dout_rows: zeros, not read
  } else {
    stage_tile(T0, base + a.H * HD, ld, a.T, tid);
    stage_tile(T1, base + 2 * a.H * HD, ld, a.T, tid);
  }
  {
    const int row = tid >> 1, half = tid & 1;     // 448 threads = 224 rows x 2 halves
    float d = 0.f;
    if (row < a.dout_rows) {
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
      s_del[row] = -d;                          // the INITIAL ACCUMULATOR of the dP product: the MFMA chain ends in dP - delta
      // lse + 3: probabilities come out pre-multiplied by scale = 2^-3 (exact in binary floating point), so that
      // dS = (P scale) (dP - delta) is ONE multiply per score; dV, which sums P scale, is multiplied by 8 at the store (exact)
      s_lse[row] = row < a.T ? a.lse2[((size_t)b * a.H + h) * a.T + row] + 3.f : 3.f;
    }
  }
  __syncthreads();
  const float c2 = a.scale * LOG2E;
  static_assert(HD == 64, "scale = 2^-3 is folded into the exponent offset");
  // The block loops are ONE basic block each (a wave-uniform `break` inside the body split it, and the scheduler no longer interleaved
  // the LDS reads, the MFMAs and the exp2 chain of an iteration: pass 1 went from 22 to 40 us): full 32-row blocks run the plain body,
  // the last, partial block runs a peeled copy that masks padded keys (pass 2) and drops a half that holds padded rows only.
  const int nfull = a.T >> 5, tail = a.T & 31;   // 6 full blocks + 5 rows at T = 197
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  typedef std::integral_constant<int, 0> Full;
  typedef std::integral_constant<int, 1> TailBoth;
  typedef std::integral_constant<int, 2> TailHalf;

  if (pass == 0) {
    // ---------------- pass 1: dK, dV for keys [32w, 32w+32), 16 keys at a time ----------------
    // Per score the vector work is one FMA, one exp2 and one multiply: padded Q / dO rows are ZERO in LDS (their P dO and dS Q
    // terms are exact zeros), `- delta` is the initial accumulator of the dP chain, the factor `scale` rides on the probability.
#pragma unroll 1
    for (int kt = 0; kt < 2; ++kt) {
      const int key0 = 32 * w + 16 * kt;
      if (key0 >= a.T || ATTN_DBG(a, 1)) break;         // wave-uniform
      const int key = key0 + l15;
      bf16x8 kf[2], vf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = global_row_frag(base + a.H * HD, ld, key, a.T, ks, lg);
        vf[ks] = global_row_frag(base + 2 * a.H * HD, ld, key, a.T, ks, lg);
      }
      f32x4 dv[4] = {zero4, zero4, zero4, zero4}, dk[4] = {zero4, zero4, zero4, zero4};
      auto block = [&](int qb, auto mode) {
        constexpr int NQT = decltype(mode)::value == 2 ? 1 : 2;     // TailHalf: the second 16 queries are padding (P dO = dS Q = 0)
        f32x4 p[2] = {zero4, zero4}, ds[2] = {zero4, zero4};       // rows q = 32qb + 16qt + 4lg + r, col key
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          const int qr = 32 * qb + 16 * qt + l15;
          const float4 del4 = *(const float4*)(s_del + 32 * qb + 16 * qt + 4 * lg);
          const f32x4 nd = {del4.x, del4.y, del4.z, del4.w};
          const f32x4 s = mfma16(row_frag(T0, qr, 1, lg), kf[1], mfma16(row_frag(T0, qr, 0, lg), kf[0], zero4));
          const f32x4 dp = mfma16(row_frag(T1, qr, 1, lg), vf[1], mfma16(row_frag(T1, qr, 0, lg), vf[0], nd));
          const float4 lse4 = *(const float4*)(s_lse + 32 * qb + 16 * qt + 4 * lg);
          const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_r[r]));     // P scale
            p[qt][r] = pr;
            ds[qt][r] = pr * dp[r];
          }
        }
        const bf16x8 pf = pack8(p[0], p[1]), dsf = pack8(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = mfma16(col_frag(T1, 32 * qb, dt, l15, lg), pf, dv[dt]);            // dV^T[d][key] (x scale); rows = d, slots = queries
          dk[dt] = mfma16(col_frag(T0, 32 * qb, dt, l15, lg), dsf, dk[dt]);           // dK^T[d][key]
        }
      };
#pragma unroll 1
      for (int qb = 0; qb < nfull; ++qb) block(qb, Full());
      if (tail > 16) block(nfull, Full());
      else if (tail > 0) block(nfull, TailHalf());
      // (v_permlane16_swap pairs lanes l and l ^ 16, which hold the SAME key: both inside or both outside the `key < T` branch)
      bf16* dst = a.dqkv + ((size_t)b * a.T + (key < a.T ? key : a.T - 1)) * ld + h * HD;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f32x4 v0 = dv[2 * k], v1 = dv[2 * k + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v0[r] *= 8.f; v1[r] *= 8.f; }      // dV summed P scale: x 8, exact
        if (key < a.T) {
          store_tile_pair(dst + a.H * HD, dk[2 * k], dk[2 * k + 1], k, lg);
          store_tile_pair(dst + 2 * a.H * HD, v0, v1, k, lg);
        }
      }
    }
  }
  // pass 2's own Q / dO row fragments: the restage form takes them from the LDS tiles it is about to overwrite (round 4, first version: from
  // global memory again -- PMC traffic 236 MB per launch against 155.5 algorithmic, half of the excess these 50 KB per item)
  bf16x8 qf[2][2], gf[2][2];
  if (!SPLIT) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[qt][ks] = row_frag(T0, 32 * w + 16 * qt + l15, ks, lg);
        gf[qt][ks] = row_frag(T1, 32 * w + 16 * qt + l15, ks, lg);
      }
    __syncthreads();                               // every wave is done with the Q / dO tiles
    stage_tile(T0, base + a.H * HD, ld, a.T, tid);
    stage_tile(T1, base + 2 * a.H * HD, ld, a.T, tid);
    __syncthreads();
  }
  if (!SPLIT || pass == 1) {
    // ---------------- pass 2: dQ for queries [32w, 32w+32), both 16-query sub-tiles in registers ----------------
    if (32 * w >= a.T || ATTN_DBG(a, 2)) return;       // wave-uniform; no barrier follows
    float lq[2];
    f32x4 ndq[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      lq[qt] = s_lse[qr];
      const float nd = s_del[qr];
      ndq[qt] = (f32x4){nd, nd, nd, nd};
      if (SPLIT) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[qt][ks] = global_row_frag(base, ld, qr, a.T, ks, lg);
          gf[qt][ks] = global_row_frag(gbase, ldo, qr, a.dout_rows, ks, lg);
        }
      }
    }
    f32x4 dq[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = zero4;
    auto block = [&](int kb, auto mode, auto nq) {
      constexpr int MODE = decltype(mode)::value, NKT = MODE == 2 ? 1 : 2, NQT = decltype(nq)::value;
      f32x4 ds[2][2] = {{zero4, zero4}, {zero4, zero4}};       // [kt][qt]: rows key = 32kb + 16kt + 4lg + r, col q
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int kr = 32 * kb + 16 * kt + l15;
        const bf16x8 k0 = row_frag(T0, kr, 0, lg), k1 = row_frag(T0, kr, 1, lg);
        const bf16x8 v0 = row_frag(T1, kr, 0, lg), v1 = row_frag(T1, kr, 1, lg);
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          const f32x4 s = mfma16(k1, qf[qt][1], mfma16(k0, qf[qt][0], zero4));
          const f32x4 dp = mfma16(v1, gf[qt][1], mfma16(v0, gf[qt][0], ndq[qt]));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lq[qt])) * dp[r];
            // last block only: a padded key has score 0 and dP - delta = -delta, and only its zero K row cancelled
            // exp2(-lse) (-delta) -- which is Inf 0 = NaN once lse < -125
            if (MODE != 0 && 32 * kb + 16 * kt + 4 * lg + r >= a.T) v = 0.f;
            ds[kt][qt][r] = v;
          }
        }
      }
      bf16x8 dsf[2];
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) dsf[qt] = pack8(ds[0][qt], ds[1][qt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kT = col_frag(T0, 32 * kb, dt, l15, lg);                        // rows = d, slots = keys
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) dq[dt][qt] = mfma16(kT, dsf[qt], dq[dt][qt]);   // dQ^T[d][q]
      }
    };
    auto sweep = [&](auto nq) {
#pragma unroll 1
      for (int kb = 0; kb < nfull; ++kb) block(kb, Full(), nq);
      if (tail > 16) block(nfull, TailBoth(), nq);
      else if (tail > 0) block(nfull, TailHalf(), nq);
    };
    if (32 * w + 16 < a.T) sweep(std::integral_constant<int, 2>());
    else sweep(std::integral_constant<int, 1>());         // the wave's second sub-tile holds padded queries only
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      bf16* dst = a.dqkv + ((size_t)b * a.T + (qr < a.T ? qr : a.T - 1)) * ld + h * HD;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (qr < a.T) store_tile_pair(dst, dq[2 * k][qt], dq[2 * k + 1][qt], k, lg);
    }
  }
}

#ifdef ROVIT_DEV   // round 3's backward (software-pipelined passes, four LDS tiles, one workgroup per CU): A/B in the developer build only
// Backward.  Pass 1: wave w owns keys [32w, 32w+32) -> dK, dV (loop over query blocks).
//            Pass 2: wave w owns queries [32w, 32w+32) -> dQ (loop over key blocks).
// Each pass recomputes the probabilities it needs from Q, K and lse2 in the orientation that makes them the
// next MFMA's operand without any data movement, so there is no cross-wave reduction and no LDS traffic
// other than operand reads.
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel_r3(const AttnArgs a) {
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
      s_del[row] = -d;                          // the INITIAL ACCUMULATOR of the dP product: the MFMA chain ends in dP - delta
      // lse + 3: probabilities come out pre-multiplied by scale = 2^-3 (exact in binary floating point), so that
      // dS = (P scale) (dP - delta) is ONE multiply per score; dV, which sums P scale, is multiplied by 8 at the store (exact)
      s_lse[row] = row < a.T ? a.lse2[((size_t)b * a.H + h) * a.T + row] + 3.f : 3.f;
    }
  }
  __syncthreads();
  const float c2 = a.scale * LOG2E;
  // Where the 56-58 us go (ablations with rovit_set_attn_debug, tools/bench_attn.py; three workgroup rounds per launch): tile
  // staging + statistics 19 us -- each round's 256 workgroups pull their 150 KB at the ~24 GB/s a CU gets on HBM misses, which
  // is also ~6 TB/s chip-wide --, pass 1 25 us (of it: S / dP products and the vector step 6, the dV / dK products with their
  // transposed operand reads 10, the dK / dV stores 6: all CUs store at once, 12.7 MB per round at the HBM write rate), pass 2
  // 13 us.  Load, compute and store bursts of the whole chip are in phase, so HBM idles while the passes run.  Tried on top of
  // this and dropped (no gain, 56-60 us): K / V tiles requested in front of pass 1 and written to LDS behind it; the column
  // fragments of step C requested in front of step B; dK / dV / dQ leaving through a wave-private LDS patch as whole 128-byte rows.
  // Round 3 (late): both passes are SOFTWARE-PIPELINED inside the wave and their vector work is cut to the minimum.  The counters
  // (profiles/r03_pmc_sq.json) showed the kernel as the SUM of its vector time (2 220 vector instructions per wave, 224 of them
  // quarter-rate exponentials: ~29 us over the three workgroup rounds) and its matrix time (392 MFMAs per wave: ~16 us), with
  // only 28 % of the matrix cycles overlapped.  Now
  //   * step A of block i+1 (the S and dP products: 16 MFMAs) is issued BEFORE the vector step B of block i (exp2 and dS of 16
  //     scores per lane) and the products C of block i (dV, dK / dQ), so every iteration holds matrix work that does not depend
  //     on its vector work;
  //   * per score the vector work is one FMA, one exp2 and one multiply: the key masks are gone (padded K / V / Q / dO rows are
  //     ZERO in LDS, so whatever a padded score is, its products are exact zeros or land in dK / dV rows that are never stored),
  //     `- delta` is the initial accumulator of the dP chain, and the factor `scale` rides on the probability (exact power of two).
  static_assert(HD == 64, "scale = 2^-3 is folded into the exponent offset");

  // ---------------- pass 1: dK, dV for keys [32w, 32w+32) ----------------
  if (!ATTN_DBG(a, 1)) {
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
    f32x4 sb[2][2][2], dpb[2][2][2];      // [buffer][qt][kt]: rows q = 32qb + 16qt + 4lg + r, col key = 32w + 16kt + l15
    auto stepA = [&](int qb, f32x4 (&s)[2][2], f32x4 (&dp)[2][2]) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qr = 32 * qb + 16 * qt + l15;
        const bf16x8 q0 = row_frag(Qs, qr, 0, lg), q1 = row_frag(Qs, qr, 1, lg);
        const bf16x8 g0 = row_frag(Gs, qr, 0, lg), g1 = row_frag(Gs, qr, 1, lg);
        const float4 del4 = *(const float4*)(s_del + 32 * qb + 16 * qt + 4 * lg);
        const f32x4 nd = {del4.x, del4.y, del4.z, del4.w};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          z = mfma16(q0, kf[kt][0], z);
          s[qt][kt] = mfma16(q1, kf[kt][1], z);
          f32x4 d = mfma16(g0, vf[kt][0], nd);
          dp[qt][kt] = mfma16(g1, vf[kt][1], d);
        }
      }
    };
    stepA(0, sb[0], dpb[0]);
#pragma unroll
    for (int qb = 0; qb < 7; ++qb) {
      TrFrag cg[4], cq[4];                       // operands of the dV / dK step: requested now, used behind the vector step
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { cg[dt] = tr_issue(Gs, 32 * qb, dt, l15, lg); cq[dt] = tr_issue(Qs, 32 * qb, dt, l15, lg); }
      if (qb + 1 < 7) stepA(qb + 1, sb[(qb + 1) & 1], dpb[(qb + 1) & 1]);
      f32x4 p[2][2], ds[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float4 lse4 = *(const float4*)(s_lse + 32 * qb + 16 * qt + 4 * lg);
        const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(fmaf(sb[qb & 1][qt][kt][r], c2, -lse_r[r]));     // P scale
            p[qt][kt][r] = pr;
            ds[qt][kt][r] = pr * dpb[qb & 1][qt][kt][r];
          }
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { pf[kt] = pack8(p[0][kt], p[1][kt]); dsf[kt] = pack8(ds[0][kt], ds[1][kt]); }
      tr_wait();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 gT = tr_val(cg[dt]);                         // rows = d, slots = queries
        const bf16x8 qT = tr_val(cq[dt]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          dv[dt][kt] = mfma16(gT, pf[kt], dv[dt][kt]);            // dV^T[d][key] (x scale)
          dk[dt][kt] = mfma16(qT, dsf[kt], dk[dt][kt]);           // dK^T[d][key]
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = 32 * w + 16 * kt + l15;
      if (key < a.T) {
        bf16* dst = a.dqkv + ((size_t)b * a.T + key) * ld + h * HD + 8 * lg;
#pragma unroll
        for (int k = 0; k < 2; ++k) {             // tile pair (2k, 2k+1) = head-dim values 32k + 8lg .. +7 (see tr_issue)
          f32x4 v0 = dv[2 * k][kt], v1 = dv[2 * k + 1][kt];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] *= 8.f; v1[r] *= 8.f; }      // dV summed P scale: x 8, exact
          *(bf16x8*)(dst + a.H * HD + 32 * k) = pack8(dk[2 * k][kt], dk[2 * k + 1][kt]);
          *(bf16x8*)(dst + 2 * a.H * HD + 32 * k) = pack8(v0, v1);
        }
      }
    }
  }

  // ---------------- pass 2: dQ for queries [32w, 32w+32) ----------------
  if (!ATTN_DBG(a, 2)) {
    bf16x8 qf[2][2], gf[2][2];
    float lq[2];
    f32x4 ndq[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      lq[qt] = s_lse[qr];
      const float nd = s_del[qr];
      ndq[qt] = (f32x4){nd, nd, nd, nd};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[qt][ks] = row_frag(Qs, qr, ks, lg); gf[qt][ks] = row_frag(Gs, qr, ks, lg); }
    }
    f32x4 dq[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 sb[2][2][2], dpb[2][2][2];      // [buffer][kt][qt]: rows key = 32kb + 16kt + 4lg + r, col q = 32w + 16qt + l15
    auto stepA = [&](int kb, f32x4 (&s)[2][2], f32x4 (&dp)[2][2]) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kr = 32 * kb + 16 * kt + l15;
        const bf16x8 k0 = row_frag(Ks, kr, 0, lg), k1 = row_frag(Ks, kr, 1, lg);
        const bf16x8 v0 = row_frag(Vs, kr, 0, lg), v1 = row_frag(Vs, kr, 1, lg);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          z = mfma16(k0, qf[qt][0], z);
          s[kt][qt] = mfma16(k1, qf[qt][1], z);
          f32x4 d = mfma16(v0, gf[qt][0], ndq[qt]);
          dp[kt][qt] = mfma16(v1, gf[qt][1], d);
        }
      }
    };
    stepA(0, sb[0], dpb[0]);
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
      TrFrag ck[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ck[dt] = tr_issue(Ks, 32 * kb, dt, l15, lg);
      if (kb + 1 < 7) stepA(kb + 1, sb[(kb + 1) & 1], dpb[(kb + 1) & 1]);
      f32x4 ds[2][2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ds[kt][qt][r] = __builtin_amdgcn_exp2f(fmaf(sb[kb & 1][kt][qt][r], c2, -lq[qt])) * dpb[kb & 1][kt][qt][r];
      bf16x8 dsf[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dsf[qt] = pack8(ds[0][qt], ds[1][qt]);
      tr_wait();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kT = tr_val(ck[dt]);                         // rows = d, slots = keys
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = mfma16(kT, dsf[qt], dq[dt][qt]);   // dQ^T[d][q]
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      if (qr < a.T) {
        bf16* dst = a.dqkv + ((size_t)b * a.T + qr) * ld + h * HD + 8 * lg;
#pragma unroll
        for (int k = 0; k < 2; ++k) *(bf16x8*)(dst + 32 * k) = pack8(dq[2 * k][qt], dq[2 * k + 1][qt]);
      }
    }
  }
}

#endif  // ROVIT_DEV

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

namespace {
// ---- the LAST block's attention: only the class token's query is ever consumed (round 4) -------------------------------------------
// Behind the last block's attention the model reads token 0 only (final norm + heads), and proj / MLP are row-wise: rovit_vit_forward
// already runs that half on the B class-token rows (round 1).  The attention itself is row-wise in the QUERY: the class token's output
// needs its own query and every key / value, nothing else -- so the last block's forward is 197 scores per (image, head) instead of
// 197 x 197, and its backward a rank-one update (dV = p (x) dO, dK = scale dS (x) q, dQ_cls = scale sum_k dS_k K_k; every other dQ row
// is exactly zero).  Exact dead-code elimination: 24 -> ~10 us forward, 44 -> ~20 us backward, both now bound by reading K / V once
// (and writing dqkv once).  fp32 FMAs on the bf16 operands (no matrix work worth the name: 50 KFLOP per item); the probabilities stay
// fp32 (the full kernel rounds them to bf16 for the P V product).  One wave per (image, head), 8 lanes per key row (16 bytes each: 128-byte
// runs), 8 keys per step.
constexpr int CLS_STEPS = TP / 8;          // 28 steps of 8 keys
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const AttnArgs a, int n_items) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= n_items) return;                                   // whole wave; no barrier in this kernel
  const int b = bh / a.H, h = bh - b * a.H;
  const int ld = 3 * a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  const int part = lane & 7, ks = lane >> 3;
  const float c2 = a.scale * LOG2E;
  float q[8];
  {
    const bf16x8 qv = *(const bf16x8*)(base + part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = (float)qv[e] * c2;
  }
  float sc[CLS_STEPS];
  float m = -INFINITY;
  // every K and V row of the item is requested up front (one wave per SIMD: 224 registers of operands are affordable): the kernel is
  // one round trip to memory, not two with the softmax between them
  bf16x8 kreg[CLS_STEPS], vreg[CLS_STEPS];
#pragma unroll
  for (int it = 0; it < CLS_STEPS; ++it) {
    const int key = 8 * it + ks, kc = key < a.T ? key : a.T - 1;
    kreg[it] = *(const bf16x8*)(base + a.H * HD + (size_t)kc * ld + part * 8);
  }
#pragma unroll
  for (int it = 0; it < CLS_STEPS; ++it) {
    const int key = 8 * it + ks, kc = key < a.T ? key : a.T - 1;       // padded keys carry p = 0
    vreg[it] = *(const bf16x8*)(base + 2 * a.H * HD + (size_t)kc * ld + part * 8);
  }
#pragma unroll
  for (int it = 0; it < CLS_STEPS; ++it) {
    const int key = 8 * it + ks;
    const bf16x8 kv = kreg[it];
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d = fmaf(q[e], (float)kv[e], d);
    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
    sc[it] = key < a.T ? d : -INFINITY;
    m = fmaxf(m, sc[it]);
  }
  m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int it = 0; it < CLS_STEPS; ++it) { sc[it] = __builtin_amdgcn_exp2f(sc[it] - m); l += sc[it]; }
  l += __shfl_xor(l, 8); l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
  if (lane == 0 && a.lse2) a.lse2[((size_t)b * a.H + h) * a.T] = m + log2f(l);
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < CLS_STEPS; ++it) {
    const bf16x8 vv = vreg[it];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(sc[it], (float)vv[e], o[e]);
  }
  const float inv_l = 1.f / l;
  bf16x8 ov;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = o[e];
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    ov[e] = (bf16)(v * inv_l);
  }
  if (ks == 0) *(bf16x8*)(a.out + (size_t)b * a.T * (a.H * HD) + h * HD + part * 8) = ov;
}

__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const AttnArgs a, int n_items) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= n_items) return;
  const int b = bh / a.H, h = bh - b * a.H;
  const int ld = 3 * a.H * HD, ldo = a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  bf16* dbase = a.dqkv + (size_t)b * a.T * ld + h * HD;
  const int part = lane & 7, ks = lane >> 3;
  const float c2 = a.scale * LOG2E;
  float q[8], g[8];
  float delta = 0.f;
  {
    const bf16x8 qv = *(const bf16x8*)(base + part * 8);
    const bf16x8 gv = *(const bf16x8*)(a.dout + (size_t)b * a.T * ldo + h * HD + part * 8);
    const bf16x8 ov = *(const bf16x8*)(a.out + (size_t)b * a.T * ldo + h * HD + part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { q[e] = (float)qv[e]; g[e] = (float)gv[e]; delta = fmaf(g[e], (float)ov[e], delta); }
  }
  delta += __shfl_xor(delta, 1); delta += __shfl_xor(delta, 2); delta += __shfl_xor(delta, 4);
  const float lse = a.lse2[((size_t)b * a.H + h) * a.T];
  float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {};
#pragma unroll 4
  for (int it = 0; it < CLS_STEPS; ++it) {
    const int key = 8 * it + ks, kc = key < a.T ? key : a.T - 1;
    const bf16x8 kv = *(const bf16x8*)(base + a.H * HD + (size_t)kc * ld + part * 8);
    const bf16x8 vv = *(const bf16x8*)(base + 2 * a.H * HD + (size_t)kc * ld + part * 8);
    float sd = 0.f, dp = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sd = fmaf(q[e], (float)kv[e], sd); dp = fmaf(g[e], (float)vv[e], dp); }
    sd += __shfl_xor(sd, 1); sd += __shfl_xor(sd, 2); sd += __shfl_xor(sd, 4);
    dp += __shfl_xor(dp, 1); dp += __shfl_xor(dp, 2); dp += __shfl_xor(dp, 4);
    const float p = key < a.T ? __builtin_amdgcn_exp2f(fmaf(sd, c2, -lse)) : 0.f;
    const float ds = p * (dp - delta);
    bf16x8 dkv, dvv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dq[e] = fmaf(ds, (float)kv[e], dq[e]);
      dkv[e] = (bf16)(a.scale * ds * q[e]);
      dvv[e] = (bf16)(p * g[e]);
    }
    if (key < a.T) {
      bf16* row = dbase + (size_t)key * ld + part * 8;
      *(bf16x8*)(row + a.H * HD) = dkv;
      *(bf16x8*)(row + 2 * a.H * HD) = dvv;
      if (key > 0) *(bf16x8*)row = zero8;          // every query but the class token's: its gradient is exactly zero (the buffer is reused)
    }
  }
  bf16x8 dqv;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = dq[e];
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    dqv[e] = (bf16)(a.scale * v);
  }
  if (ks == 0) *(bf16x8*)(dbase + part * 8) = dqv;
}

}  // namespace

// the last block's attention when only the class token's row of `out` is consumed / of `dout` carries gradient (include/rovit_hip.h).
// Writes out[b, 0, :] and lse2[b, h, 0] only; the backward writes the whole dqkv (dQ rows 1.. are zeros).
extern "C" int rovit_attention_cls_fwd(const void* qkv, void* out, float* lse2, int batch, int tokens, int heads, int head_dim, float scale,
                                       rovit_stream_t stream) {
  ROVIT_CHECK_ARG(qkv && out, ROVIT_ERR_NULL, "attention_cls_fwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && tokens > 0 && tokens <= TP && head_dim == HD && heads > 0, ROVIT_ERR_SHAPE, "attention_cls_fwd: unsupported shape");
  ROVIT_CHECK_ARG(rovit_aligned16(qkv) && rovit_aligned16(out), ROVIT_ERR_ALIGN, "attention_cls_fwd: alignment");
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.lse2 = lse2; a.T = tokens; a.H = heads; a.scale = scale;
  const int items = batch * heads;
  hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, items);
  ROVIT_CHECK_LAUNCH("attn_cls_fwd_kernel");
  return ROVIT_OK;
}
extern "C" int rovit_attention_cls_bwd(const void* qkv, const void* out, const float* lse2, const void* dout, void* dqkv, int batch, int tokens,
                                       int heads, int head_dim, float scale, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(qkv && out && lse2 && dout && dqkv, ROVIT_ERR_NULL, "attention_cls_bwd: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && tokens > 0 && tokens <= TP && head_dim == HD && heads > 0, ROVIT_ERR_SHAPE, "attention_cls_bwd: unsupported shape");
  ROVIT_CHECK_ARG(rovit_aligned16(qkv) && rovit_aligned16(out) && rovit_aligned16(dout) && rovit_aligned16(dqkv), ROVIT_ERR_ALIGN,
                  "attention_cls_bwd: alignment");
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.out = (bf16*)const_cast<void*>(out); a.lse2 = (float*)lse2; a.T = tokens; a.H = heads; a.scale = scale;
  a.dout = (const bf16*)dout; a.dqkv = (bf16*)dqkv; a.dout_rows = 1;
  const int items = batch * heads;
  hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, items);
  ROVIT_CHECK_LAUNCH("attn_cls_bwd_kernel");
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
#ifdef ROVIT_DEV
  if (ROVIT_KNOB(ROVIT_KNOB_ATTN_FWD_R3, 0)) {
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_fwd_kernel_r3, lds), ROVIT_ERR_LAUNCH, "attention_fwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(attn_fwd_kernel_r3, dim3(batch * heads), dim3(NW * 64), lds, (hipStream_t)stream, a);
    ROVIT_CHECK_LAUNCH("attn_fwd_kernel_r3");
    return ROVIT_OK;
  }
#endif
#ifdef ROVIT_DEV
  if (ROVIT_KNOB(ROVIT_KNOB_ATTN_FWD3, 0)) {       // three workgroups per CU (unpadded swizzled tiles, scores computed twice)
    const size_t lds3 = (size_t)2 * TK3 * HD * sizeof(bf16);
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_fwd3_kernel, lds3), ROVIT_ERR_LAUNCH, "attention_fwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(attn_fwd3_kernel, dim3(batch * heads), dim3(NW * 64), lds3, (hipStream_t)stream, a);
    ROVIT_CHECK_LAUNCH("attn_fwd3_kernel");
    return ROVIT_OK;
  }
#endif
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_fwd_kernel, lds), ROVIT_ERR_LAUNCH, "attention_fwd: cannot raise the LDS limit");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(batch * heads), dim3(NW * 64), lds, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("attn_fwd_kernel");
  return ROVIT_OK;
}

// internal (common.h): dout_rows = rows of every image's dout that carry gradient; rows beyond are zeros that are never read
int rovit_attention_bwd_rows(const void* qkv, const void* out, const float* lse2, const void* dout, int dout_rows, void* dqkv, int batch,
                             int tokens, int heads, int head_dim, float scale, rovit_stream_t stream);
extern "C" int rovit_attention_bwd(const void* qkv, const void* out, const float* lse2, const void* dout, void* dqkv, int batch,
                                   int tokens, int heads, int head_dim, float scale, rovit_stream_t stream) {
  return rovit_attention_bwd_rows(qkv, out, lse2, dout, tokens, dqkv, batch, tokens, heads, head_dim, scale, stream);
}
int rovit_attention_bwd_rows(const void* qkv, const void* out, const float* lse2, const void* dout, int dout_rows, void* dqkv, int batch,
                             int tokens, int heads, int head_dim, float scale, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(qkv && out && lse2 && dout && dqkv, ROVIT_ERR_NULL, "attention_bwd: null pointer");
  int rc = check_attn(batch, tokens, heads, head_dim);
  if (rc) return rc;
  ROVIT_CHECK_ARG(rovit_aligned16(qkv) && rovit_aligned16(out) && rovit_aligned16(dout) && rovit_aligned16(dqkv), ROVIT_ERR_ALIGN,
                  "attention_bwd: alignment");
  AttnArgs a{};
  a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.lse2 = (float*)lse2; a.T = tokens; a.H = heads; a.scale = scale;
  a.dout = (const bf16*)dout; a.dqkv = (bf16*)dqkv;
  ROVIT_CHECK_ARG(dout_rows >= 1 && dout_rows <= tokens, ROVIT_ERR_SHAPE, "attention_bwd: dout_rows out of range");
  a.dout_rows = dout_rows;
#ifdef ROVIT_DEV
  a.dbg = ROVIT_KNOB(ROVIT_KNOB_ATTN_DBG, 0);
#endif
#ifdef ROVIT_DEV
  if (ROVIT_KNOB(ROVIT_KNOB_ATTN_BWD_R3, 0)) {
    const size_t lds3 = (size_t)4 * TP * AST * sizeof(bf16) + 2 * TP * sizeof(float);
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_kernel_r3, lds3), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(attn_bwd_kernel_r3, dim3(batch * heads), dim3(NW * 64), lds3, (hipStream_t)stream, a);
    ROVIT_CHECK_LAUNCH("attn_bwd_kernel_r3");
    return ROVIT_OK;
  }
#endif
  const size_t lds = (size_t)2 * TP * AST * sizeof(bf16) + 2 * TP * sizeof(float);
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_kernel<false>, lds), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
  const int items = batch * heads;
#ifdef ROVIT_DEV
  if (ROVIT_KNOB(ROVIT_KNOB_ATTN_BWD_SPLIT, 0)) {
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_kernel<true>, lds), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
    const int groups = ((items + 7) / 8 + 31) / 32;          // groups of 32 items per XCD label; 2 x 32 x 8 workgroups each
    hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(groups * 512), dim3(NW * 64), lds, (hipStream_t)stream, a, items);
    ROVIT_CHECK_LAUNCH("attn_bwd_kernel<split>");
    return ROVIT_OK;
  }
#endif
  hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(items), dim3(NW * 64), lds, (hipStream_t)stream, a, items);
  ROVIT_CHECK_LAUNCH("attn_bwd_kernel");
  return ROVIT_OK;
}
