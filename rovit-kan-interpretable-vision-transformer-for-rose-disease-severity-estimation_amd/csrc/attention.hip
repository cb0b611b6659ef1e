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
  int dbg;              // developer knob (timing ablations, rovit_set_attn_debug): bit 0 skip pass 1, bit 1 skip pass 2, bit 2 skip exp2
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
__device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 tr_val(const TrFrag& f) { return cat4(f.lo, f.hi); }

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
  if (!(a.dbg & 1)) {                            // (developer knob rovit_set_attn_debug: timing ablations)
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
  if (!(a.dbg & 2)) {
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

// ------------------------------------------------------------------------------------------------------------------
// Backward, late round 3: ONE pass over the scores ("ring").  The two-pass kernel above computes S, P and dP twice, once in each
// orientation, because dK / dV contract over the queries and dQ over the keys.  Here every wave owns 32 keys for the whole
// launch (their K / V row fragments AND the transposed K fragments live in registers) and walks the seven 32-query blocks in a
// ROTATED order -- in step t wave w works on query block (w + t) mod 7 -- so that in every step the seven waves hold seven
// different query blocks.  Per step and wave: S and dP of (32 queries x 32 keys) once, one exp2 per score, dV and dK accumulated in
// registers as before, and the block's dQ contribution added to a fp32 dQ tile in LDS that only this wave touches in this step
// (one barrier per step hands the tiles on; the order of additions to a tile is fixed: wave (qb - t) mod 7 at step t, so the result
// is bit-reproducible).  dS reaches the dQ product through a wave-private 32 x 32 bf16 patch in LDS (written in the accumulator
// layout, read back transposed with ds_read_b64_tr_b16): 5 MFMA products and one exp2 per score instead of 7 and two.
// LDS: Q and dO tiles (72 KB), the dQ tile (224 x 68 fp32 = 61 KB; the K tile is staged THERE first, only to be read back
// transposed into registers), statistics, patches: 156 KB.  K and V tiles are never needed.
// MEASURED (MI355X, batch 256): correct (tests/test_gpu_round3.py, every shape of the two-pass kernels' test) and SLOWER, 71 us
// against 57-58 us, so it is opt-in (rovit_set_attn_bwd_pipe(2) / ROVIT_ATTN_BWD_PIPE=2).  Per step and CU the work is 280 MFMAs (1 280
// matrix cycles per SIMD for its two waves), ~900 vector cycles and ~1 700 LDS cycles (the dQ read-modify-write alone: 56
// ds_write_b128 at 13 cycles + 56 ds_read_b128; the patch; 140 transposed reads), and the barrier that hands the dQ tiles on keeps
// the seven waves in LOCKSTEP, so these add up (2.4 us per step) instead of overlapping as they do between the free-running waves
// of the two-pass kernel.  Fewer operations, worse overlap: the two-pass kernel stays the default.
constexpr int DQ_ST = HD + 4;                  // fp32 row stride of the dQ tile (272 bytes)
constexpr int DS_ST = 48;                      // bf16 row stride of a wave's dS patch (96 bytes: an odd multiple of 32)
constexpr size_t ATTN_RING_LDS = (size_t)2 * TP * AST * sizeof(bf16) + (size_t)TP * DQ_ST * sizeof(float) + 2 * TP * sizeof(float) +
                                 (size_t)NW * 32 * DS_ST * sizeof(bf16);
static_assert((size_t)TP * AST * sizeof(bf16) <= (size_t)TP * DQ_ST * sizeof(float), "the K tile is staged inside the dQ tile");
__device__ __forceinline__ bf16x8 col_frag_s(const bf16* tile, int stride, int r0, int dt, int l15, int lg) {
  const bf16* p = tile + (r0 + 4 * lg + (l15 >> 2)) * stride + dt * 16 + 4 * (l15 & 3);
  return cat4(lds_read_tr(p), lds_read_tr(p + 16 * stride));
}

__global__ __launch_bounds__(NW * 64) void attn_bwd_ring_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  bf16* Qs = lds;
  bf16* Gs = Qs + TP * AST;                     // dO
  float* dQa = (float*)(Gs + TP * AST);         // [TP][DQ_ST]
  float* s_lse = dQa + TP * DQ_ST;              // [TP]
  float* s_del = s_lse + TP;                    // [TP]
  bf16* Ka = (bf16*)dQa;                        // K tile [TP][AST], only until the transposed fragments are in registers
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  bf16* patch = (bf16*)(s_del + TP) + w * 32 * DS_ST;
  const int ld = 3 * a.H * HD, ldo = a.H * HD;
  const bf16* base = a.qkv + (size_t)b * a.T * ld + h * HD;
  const bf16* gbase = a.dout + (size_t)b * a.T * ldo + h * HD;
  const bf16* obase = a.out + (size_t)b * a.T * ldo + h * HD;
  stage_tile(Qs, base, ld, a.T, tid);
  stage_tile(Gs, gbase, ldo, a.T, tid);
  stage_tile(Ka, base + a.H * HD, ld, a.T, tid);
  bf16x8 kf[2][2], vf[2][2];                    // this wave's 32 keys as MFMA operands (zero rows beyond T)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int kr = 32 * w + 16 * kt + l15;
    const int kc = kr < a.T ? kr : a.T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[kt][ks] = keep_if(*(const bf16x8*)(base + a.H * HD + (size_t)kc * ld + ks * 32 + lg * 8), kr < a.T);
      vf[kt][ks] = keep_if(*(const bf16x8*)(base + 2 * a.H * HD + (size_t)kc * ld + ks * 32 + lg * 8), kr < a.T);
    }
  }
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
      s_del[row] = -d;                          // initial accumulator of the dP chain (see attn_bwd_kernel)
      s_lse[row] = row < a.T ? a.lse2[((size_t)b * a.H + h) * a.T + row] + 3.f : 3.f;      // + 3: P comes out times scale = 2^-3
    }
  }
  __syncthreads();
  bf16x8 kT[4];                                 // K^T of the wave's keys: rows = d, contraction slots = the 32 keys
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) kT[dt] = col_frag(Ka, 32 * w, dt, l15, lg);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                               // nobody reads the K tile any more: its space is the dQ tile from here on
  for (int i = tid; i < TP * DQ_ST / 4; i += NW * 64) ((f32x4*)dQa)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};      // (every step adds: no first-visitor branch)
  __syncthreads();
  const float c2 = a.scale * LOG2E;
  static_assert(HD == 64, "scale = 2^-3 is folded into the exponent offset");
  f32x4 dv[4][2], dk[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
  for (int t = 0; t < 7; ++t) {
    const int qb = w + t >= 7 ? w + t - 7 : w + t;
    // ---- S and dP - delta: rows q = 32qb + 16qt + 4lg + r, col key = 32w + 16kt + l15 ----
    f32x4 p[2][2], ds[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * qb + 16 * qt + l15;
      const bf16x8 q0 = row_frag(Qs, qr, 0, lg), q1 = row_frag(Qs, qr, 1, lg);
      const bf16x8 g0 = row_frag(Gs, qr, 0, lg), g1 = row_frag(Gs, qr, 1, lg);
      const float4 del4 = *(const float4*)(s_del + 32 * qb + 16 * qt + 4 * lg);
      const float4 lse4 = *(const float4*)(s_lse + 32 * qb + 16 * qt + 4 * lg);
      const f32x4 nd = {del4.x, del4.y, del4.z, del4.w};
      const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = mfma16(q0, kf[kt][0], sc);
        sc = mfma16(q1, kf[kt][1], sc);
        f32x4 dp = mfma16(g0, vf[kt][0], nd);
        dp = mfma16(g1, vf[kt][1], dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -lse_r[r]));     // P scale
          p[qt][kt][r] = pr;
          ds[qt][kt][r] = pr * dp[r];                                               // scale P (dP - delta)
        }
      }
    }
    // ---- dS, transposed, for the dQ product: the wave's patch holds [key][query] ----
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) *(bf16x4*)(patch + (16 * kt + l15) * DS_ST + 16 * qt + 4 * lg) = pack4(ds[qt][kt]);
    // ---- dV, dK (contraction over the block's queries) ----
    bf16x8 pf[2], dsf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { pf[kt] = pack8(p[0][kt], p[1][kt]); dsf[kt] = pack8(ds[0][kt], ds[1][kt]); }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 gT = col_frag(Gs, 32 * qb, dt, l15, lg);       // rows = d, slots = queries
      const bf16x8 qT = col_frag(Qs, 32 * qb, dt, l15, lg);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        dv[dt][kt] = mfma16(gT, pf[kt], dv[dt][kt]);              // dV^T[d][key] (x scale)
        dk[dt][kt] = mfma16(qT, dsf[kt], dk[dt][kt]);             // dK^T[d][key]
      }
    }
    // ---- dQ^T[d][q] += K^T dS^T (contraction over the wave's 32 keys), accumulated in the block's LDS tile ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the patch writes have landed
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const bf16x8 dsT = col_frag_s(patch, DS_ST, 0, qt, l15, lg);   // cols = the tile's 16 queries, slots = keys
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float* ap = dQa + (32 * qb + 16 * qt + l15) * DQ_ST + 16 * dt + 4 * lg;
        *(f32x4*)ap = mfma16(kT[dt], dsT, *(const f32x4*)ap);
      }
    }
    __syncthreads();                                               // hand the dQ tiles (and nothing else) on
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = 32 * w + 16 * kt + l15;
    if (key < a.T) {
      bf16* dst = a.dqkv + ((size_t)b * a.T + key) * ld + h * HD + 4 * lg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 v = dv[dt][kt];
        v[0] *= 8.f; v[1] *= 8.f; v[2] *= 8.f; v[3] *= 8.f;
        *(bf16x4*)(dst + a.H * HD + 16 * dt) = pack4(dk[dt][kt]);
        *(bf16x4*)(dst + 2 * a.H * HD + 16 * dt) = pack4(v);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qr = 32 * w + 16 * qt + l15;
    if (qr < a.T) {
      bf16* dst = a.dqkv + ((size_t)b * a.T + qr) * ld + h * HD + 4 * lg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(dst + 16 * dt) = pack4(*(const f32x4*)(dQa + qr * DQ_ST + 16 * dt + 4 * lg));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward, round 3: the same two passes and the same arithmetic (bit-identical dQ / dK / dV), but PERSISTENT workgroups whose
// tiles arrive by LDS-DMA (global_load_lds_dwordx4) while a pass computes.  The kernel above stages four tiles (143 KB, one
// workgroup per CU), computes, exits: an HBM-bound staging burst and a compute phase strictly alternate (62 us at batch 256
// for 154 MB; ablation: the data movement alone takes 33 us = 4.7 TB/s, the two passes 31 us, and they do not overlap).
// Here a workgroup walks its (image, head) items with
//   LDS = [lse, delta | A0 = Q,dO | A1 = Q,dO | B = K,V]   (tiles of 13 x 16 rows: 26 KB each, 158 KB in all)
//   * pass 1 of item t (keys-owner: reads Q / dO from A[t&1]; K / V fragments in registers) runs while the Q / dO tiles of item
//     t+1 land in the OTHER A buffer and the K / V tiles of item t land in B;
//   * pass 2 of item t (queries-owner: reads K / V from B; Q / dO fragments in registers) runs while the K / V row fragments,
//     lse and O rows of item t+1 arrive as plain loads (consumed behind the pass);
//   * delta = rowsum(dO * O) comes from the staged dO tile and those O rows (no O tile, no second read of dO).
// LDS-DMA writes lane-linear 1 KB pieces, so a tile is stored as [row block of 16][column half of 32][16 rows][64 bytes] with
// the 16-byte chunk x of row r at x ^ g4(r >> 2), g4 = {0,2,3,1} (swizzle on the per-lane SOURCE address and on the reads):
// conflict-free for the ds_read_b128 row fragments and the ds_read_b64_tr_b16 column fragments (tools/lds_attn_image_check.py).
// Rows beyond T are not zero-filled (a DMA cannot write zeros: the source row is clamped, and the 14th row block a 32-row
// step touches is whatever follows the tile): padded QUERIES get lse = +inf, so their probabilities are exp2(-inf) = 0
// exactly and every product they enter is an exact zero; padded KEYS are masked as before.
// Completion is hand-counted: every DMA batch (one A buffer, or B) is 8 pieces per wave (52 pieces over 7 waves, the last four
// slots re-load pieces 0-3), so `vmcnt(8)` = "everything but the batch issued last has landed".
// ------------------------------------------------------------------------------------------------------------------
constexpr int TROWS = 208;                       // rows a tile really holds (13 row blocks; T <= 208)
constexpr int TILE_E = 13 * 2 * 512;             // bf16 elements of a [208][64] tile in the DMA image (26 KB)
constexpr int TILE_PIECES = 26;
__device__ __forceinline__ int g4(int q) { return (0x1E >> (2 * q)) & 3; }          // {0,2,3,1}[q]
// element offset of the 16-byte chunk lc (0..3) of column half h of row `row`
__device__ __forceinline__ int img_off(int row, int h, int lc) {
  return ((row >> 4) * 2 + h) * 512 + (row & 15) * 32 + ((lc ^ g4((row & 15) >> 2)) * 8);
}
__device__ __forceinline__ bf16x8 row_frag_i(const bf16* tile, int row, int ks, int lg) { return *(const bf16x8*)(tile + img_off(row, ks, lg)); }
// Column fragment: rows r0 + 4 lg + (l15 >> 2) (+16), columns 16 dt + 4 (l15 & 3) .. +3; r0 a multiple of 32.
// The transposed reads are issued as inline asm: with an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 (it cannot prove that the read does not touch the tile being filled), which would drain the
// prefetch in the middle of the pass it is meant to overlap.  The asm reads are invisible to the compiler's counters, so
// col_wait() -- lgkmcnt(0) + a scheduling fence (cdna guide 5.4 rule 18) -- stands between them and the first MFMA that uses them.
struct ColFrag { bf16x4 lo, hi; };
__device__ __forceinline__ ColFrag col_frag_issue(const bf16* tile, int r0, int dt, int l15, int lg) {
  const bf16* p = tile + img_off(r0 + 4 * lg + (l15 >> 2), dt >> 1, 2 * (dt & 1) + ((l15 & 3) >> 1)) + 4 * (l15 & 1);
  const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) bf16*)p;
  ColFrag f;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(f.lo), "=&v"(f.hi) : "v"(a0) : "memory");   // + 16 rows = next row block
  return f;
}
__device__ __forceinline__ void col_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 col_val(const ColFrag& f) { return cat4(f.lo, f.hi); }

template <int N>
__device__ __forceinline__ void attn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr size_t ATTN_PIPE_LDS = (size_t)2 * TP * sizeof(float) + (size_t)6 * TILE_E * sizeof(bf16) + 2048;   // + one row block of slack behind V

__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_pipe_kernel(const AttnArgs a, int n_items) {
  // ONE array: lse / delta, then A0, A1 = [Q | dO], then B = [K | V].  NB the statistics are READ through the array's own element
  // type (bf16x8 loads bit-cast to four floats): read through a float pointer, every lse / delta read of pass 1 carried a
  // compiler-inserted vmcnt(0) (hipcc orders such a read against the pending LDS-DMA; the tile reads are exempt), which drained
  // the prefetch at the top of the pass it is meant to overlap.  Check the .s for `s_waitcnt vmcnt` after any edit here: the only
  // ones allowed are the prologue's, the hand-placed vmcnt(8)s and the consume in step 8.
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  float* s_lse = (float*)lds;
  float* s_del = s_lse + TP;
  bf16* Abuf = lds + 4 * TP;                     // 2 * TP floats
  bf16* Ks = Abuf + 4 * TILE_E;
  bf16* Vs = Ks + TILE_E;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int ld = 3 * a.H * HD, ldo = a.H * HD;
  const float c2 = a.scale * LOG2E;
  // this lane's place in a DMA piece: row r = lane >> 2 of the piece's 16 rows, physical chunk lane & 3 = logical chunk ^ g4
  const int p_r = lane >> 2, p_col = ((lane & 3) ^ g4(p_r >> 2)) * 8;

  auto item_ptrs = [&](int item, const bf16*& base, const bf16*& gbase, const bf16*& obase, int& b, int& h) {
    b = item / a.H; h = item - b * a.H;
    base = a.qkv + (size_t)b * a.T * ld + h * HD;
    gbase = a.dout + (size_t)b * a.T * ldo + h * HD;
    obase = a.out + (size_t)b * a.T * ldo + h * HD;
  };
  // One DMA batch = the two tiles of a buffer = 52 pieces; wave w issues slots w + 7 i, i < 8 (slots 52..55 re-load pieces 0..3,
  // so that every wave issues exactly 8).  (The lane-constant parts of the source addresses are made opaque per call: hoisted
  // to kernel entry, the per-piece offsets would live across both passes and spill -- and a spill reload is a vmcnt(0).)
  int p_ro = p_r, p_co = p_col;
  auto dma_pair = [&](const bf16* src0, int ld0, const bf16* src1, int ld1, bf16* dst) {
    if (a.dbg & 4) return;                                  // timing ablation: no tile traffic
    asm volatile("" : "+v"(p_ro), "+v"(p_co));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int j = w + NW * i;
      j = j >= 2 * TILE_PIECES ? j - 2 * TILE_PIECES : j;
      const bool second = j >= TILE_PIECES;                 // wave-uniform
      const int pj = second ? j - TILE_PIECES : j;
      const bf16* src = second ? src1 : src0;
      const int sld = second ? ld1 : ld0;
      int row = 16 * (pj >> 1) + p_ro;
      row = row < a.T ? row : a.T - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(row * sld + 32 * (pj & 1) + p_co)),
                                       (__attribute__((address_space(3))) void*)(dst + j * 512), 16, 0, 0);
    }
  };
  // the same, ONE slot i (0..7) of the batch: inside pass 1 the two refills are issued a piece or two per query block, because a
  // wave that issues its 8 pieces back to back sits in the vector-memory issue queue until most of them have been accepted
  // (the CU holds a bounded number of requests in flight): measured, a burst of 8 + 8 pieces per wave in front of a pass did
  // not overlap with that pass at all
  auto dma_slot = [&](const bf16* src0, int ld0, const bf16* src1, int ld1, bf16* dst, int i) {
    if (a.dbg & 4) return;
    int pr = p_r, pc = p_col;
    asm volatile("" : "+v"(pr), "+v"(pc));
    int j = w + NW * i;
    j = j >= 2 * TILE_PIECES ? j - 2 * TILE_PIECES : j;
    const bool second = j >= TILE_PIECES;                 // wave-uniform
    const int pj = second ? j - TILE_PIECES : j;
    const bf16* src = second ? src1 : src0;
    const int sld = second ? ld1 : ld0;
    int row = 16 * (pj >> 1) + pr;
    row = row < a.T ? row : a.T - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(row * sld + 32 * (pj & 1) + pc)),
                                     (__attribute__((address_space(3))) void*)(dst + j * 512), 16, 0, 0);
  };
  // K / V row fragments of this wave's 32 keys, the lse of row `tid` and the O values thread (row = tid >> 1, half = tid & 1)
  // needs for delta, straight from global memory
  auto load_next = [&](const bf16* base, const bf16* obase, int b, int h, bf16x8 (&kf)[2][2], bf16x8 (&vf)[2][2], float& lse_v, bf16x8 (&of)[4]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      int row = 32 * w + 16 * kt + l15;
      row = row < a.T ? row : a.T - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[kt][ks] = *(const bf16x8*)(base + a.H * HD + (unsigned)(row * ld + ks * 32 + lg * 8));
        vf[kt][ks] = *(const bf16x8*)(base + 2 * a.H * HD + (unsigned)(row * ld + ks * 32 + lg * 8));
      }
    }
    const int lr = tid < a.T ? tid : a.T - 1;
    lse_v = a.lse2[((size_t)b * a.H + h) * a.T + lr];
    const int orow = (tid >> 1) < a.T ? (tid >> 1) : a.T - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) of[i] = *(const bf16x8*)(obase + (unsigned)(orow * ldo + (tid & 1) * 32 + i * 8));
  };
  auto consume = [&](bf16x8 (&kf)[2][2], bf16x8 (&vf)[2][2], float& lse_v, bf16x8 (&of)[4]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(kf[kt][ks]), "+v"(vf[kt][ks]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(of[i]));
    asm volatile("" : "+v"(lse_v));
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  // A 32-row step over rows 192..223 reads one row block past a 208-row tile: the first rows of whatever follows it (the next
  // tile, or the slack behind V).  Those products are masked to exact zeros only if what is read is FINITE (0 x NaN = NaN), so
  // the whole allocation starts as zeros; afterwards it only ever holds zeros or real (finite) tile data.
  for (int e = tid; e < (int)(ATTN_PIPE_LDS / 16); e += NW * 64) ((f32x4*)lds)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  barrier_lds();
  const bf16 *base, *gbase, *obase;
  int b, h;
  item_ptrs(item, base, gbase, obase, b, h);
  dma_pair(base, ld, gbase, ldo, Abuf);                               // A0 <- Q, dO of the first item
  dma_pair(base + a.H * HD, ld, base + 2 * a.H * HD, ld, Ks);         // B  <- K, V
  bf16x8 kf[2][2], vf[2][2], of[4];
  float lse_v;
  load_next(base, obase, b, h, kf, vf, lse_v, of);
  consume(kf, vf, lse_v, of);          // one vmcnt(0) in the prologue, so that no compiler wait for these loads sits inside the loop
  __builtin_amdgcn_s_barrier();        // ... and every wave's pieces of the first item's tiles have landed
  int cur = 0;
  bool first_item = true;              // its K / V tiles came in with the prologue

  for (;;) {
    bf16* Qs = Abuf + cur * 2 * TILE_E;
    bf16* Gs = Qs + TILE_E;
    const int next = item + gridDim.x;
    const bool more = next < n_items;               // workgroup-uniform
    const bf16 *nbase = base, *ngbase = gbase, *nobase = obase;
    int nb = b, nh = h;
    if (more) item_ptrs(next, nbase, ngbase, nobase, nb, nh);
    // ---- 1. Q / dO of this item have landed: every wave passed the vmcnt(0) of step 8 (prologue: of the first consume) after
    //         issuing them and then a barrier.  No wait here: the youngest operations are the previous item's stores, and
    //         waiting for stores that were issued a moment ago is exactly what this schedule avoids ----
    asm volatile("" ::: "memory");
    // ---- 2. lse (+inf on padded queries) and delta = scale * rowsum(dO * O): dO from the staged tile, O from registers ----
    if (tid < TP) s_lse[tid] = tid < a.T ? lse_v : INFINITY;
    {
      const int row = tid >> 1, half = tid & 1;     // 448 threads = 224 rows x 2 halves
      const int rc = row < TROWS ? row : TROWS - 1;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 g = *(const bf16x8*)(Gs + img_off(rc, half, i));
#pragma unroll
        for (int q = 0; q < 8; ++q) d = fmaf((float)g[q], (float)of[i][q], d);
      }
      d += __shfl_xor(d, 1);
      if (half == 0) s_del[row] = d * a.scale;      // pre-scaled: dS = P * (dP*scale - delta*scale)
    }
    barrier_lds();
    // ---- 3. the OTHER A buffer (last read by pass 1 of the previous item) takes Q / dO of the next item; pass 1 ----
    bf16* Anext = Abuf + (cur ^ 1) * 2 * TILE_E;
    f32x4 dv[4][2], dk[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    bool kvalid[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) kvalid[kt] = (32 * w + 16 * kt + l15) < a.T;
    if (a.dbg & 1) {                                     // timing ablation without pass 1: the refills as bursts
      if (!first_item) dma_pair(base + a.H * HD, ld, base + 2 * a.H * HD, ld, Ks);
      if (more) { dma_pair(nbase, ld, ngbase, ldo, Anext); }
    }
    if (!(a.dbg & 1))
#pragma unroll 1
    for (int qb = 0; qb < 7; ++qb) {
      // refills, spread over the pass: K / V of THIS item (B, free since the previous item's pass 2; needed by pass 2) two slots per
      // block in blocks 0-3, Q / dO of the NEXT item (the other A buffer) one slot per block and the eighth with the last.
      // Issue order per wave: B0 B1 A0 | B2 B3 A1 | B4 B5 A2 | B6 B7 A3 | A4 | A5 | A6 A7
      if (qb < 4 && !first_item) {
        dma_slot(base + a.H * HD, ld, base + 2 * a.H * HD, ld, Ks, 2 * qb);
        dma_slot(base + a.H * HD, ld, base + 2 * a.H * HD, ld, Ks, 2 * qb + 1);
      }
      if (more) {
        dma_slot(nbase, ld, ngbase, ldo, Anext, qb);
        if (qb == 6) dma_slot(nbase, ld, ngbase, ldo, Anext, 7);
      }
      // the transposed operands of this query block first: their latency hides behind the S / dP products and the exponentials
      ColFrag gT[4], qT[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        gT[dt] = col_frag_issue(Gs, 32 * qb, dt, l15, lg);          // rows = d, slots = queries
        qT[dt] = col_frag_issue(Qs, 32 * qb, dt, l15, lg);
      }
      f32x4 p[2][2], ds[2][2];     // [qt][kt]: rows q = 32qb + 16qt + 4lg + r, col key = 32w + 16kt + l15
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qr = 32 * qb + 16 * qt + l15;
        const bf16x8 q0 = row_frag_i(Qs, qr, 0, lg), q1 = row_frag_i(Qs, qr, 1, lg);
        const bf16x8 g0 = row_frag_i(Gs, qr, 0, lg), g1 = row_frag_i(Gs, qr, 1, lg);
        // (read through the array's element type: see the note at the LDS layout)
        const f32x4 lse4 = __builtin_bit_cast(f32x4, *(const bf16x8*)(lds + 2 * (32 * qb + 16 * qt + 4 * lg)));
        const f32x4 del4 = __builtin_bit_cast(f32x4, *(const bf16x8*)(lds + 2 * (TP + 32 * qb + 16 * qt + 4 * lg)));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(q0, kf[kt][0], s);  s = mfma16(q1, kf[kt][1], s);
          dp = mfma16(g0, vf[kt][0], dp); dp = mfma16(g1, vf[kt][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = kvalid[kt] ? __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse4[r])) : 0.f;
            p[qt][kt][r] = pr;
            ds[qt][kt][r] = pr * fmaf(dp[r], a.scale, -del4[r]);
          }
        }
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) { pf[kt] = pack8(p[0][kt], p[1][kt]); dsf[kt] = pack8(ds[0][kt], ds[1][kt]); }
      col_wait();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          dv[dt][kt] = mfma16(col_val(gT[dt]), pf[kt], dv[dt][kt]);     // dV^T[d][key]
          dk[dt][kt] = mfma16(col_val(qT[dt]), dsf[kt], dk[dt][kt]);    // dK^T[d][key]
        }
    }
    // ---- 4. this wave's Q / dO row fragments and statistics for pass 2 ----
    bf16x8 qf[2][2], gf[2][2];
    float lq[2], dq_[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      lq[qt] = s_lse[qr]; dq_[qt] = s_del[qr];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[qt][ks] = row_frag_i(Qs, qr, ks, lg); gf[qt][ks] = row_frag_i(Gs, qr, ks, lg); }
    }
    // ---- 5. K / V of this item have landed in every wave (the 5 youngest operations are Q / dO slots of the next item; the
    //         previous item's stores are older than the K / V pieces and a whole pass 1 old by now) ----
    if (more) attn_wait_vm<5>(); else attn_wait_vm<0>();        // younger than B7: A3 .. A7
    barrier_lds();
    first_item = false;
    // ---- 6. dK / dV out (a whole pass 2 before the consume of step 8 waits for them); request the next item's K / V fragments,
    //         lse and O rows ----
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = 32 * w + 16 * kt + l15;
      if (key < a.T && !(a.dbg & 8)) {
        bf16* dst = a.dqkv + ((size_t)b * a.T + key) * ld + h * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          *(bf16x4*)(dst + a.H * HD + 16 * dt) = pack4(dk[dt][kt]);
          *(bf16x4*)(dst + 2 * a.H * HD + 16 * dt) = pack4(dv[dt][kt]);
        }
      }
    }
    bf16x8 kf2[2][2], vf2[2][2], of2[4];
    float lse2v = lse_v;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf2[kt][ks] = kf[kt][ks]; vf2[kt][ks] = vf[kt][ks]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) of2[i] = of[i];
    if (more && !(a.dbg & 4)) load_next(nbase, nobase, nb, nh, kf2, vf2, lse2v, of2);
    // ---- 7. pass 2: dQ for queries [32w, 32w+32) ----
    f32x4 dq[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(a.dbg & 2))
#pragma unroll 1
    for (int kb = 0; kb < 7; ++kb) {
      ColFrag kT[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) kT[dt] = col_frag_issue(Ks, 32 * kb, dt, l15, lg);     // rows = d, slots = keys
      f32x4 ds[2][2];             // [kt][qt]: rows key = 32kb + 16kt + 4lg + r, col q = 32w + 16qt + l15
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kr = 32 * kb + 16 * kt + l15;
        const bf16x8 k0 = row_frag_i(Ks, kr, 0, lg), k1 = row_frag_i(Ks, kr, 1, lg);
        const bf16x8 v0 = row_frag_i(Vs, kr, 0, lg), v1 = row_frag_i(Vs, kr, 1, lg);
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
      col_wait();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = mfma16(col_val(kT[dt]), dsf[qt], dq[dt][qt]);   // dQ^T[d][q]
    }
    // ---- 8. the next item's fragments are consumed HERE, unconditionally (the compiler's wait for these plain loads is a
    //         vmcnt(0): it must come before the B refill is issued, and on every path, or its wait-count analysis keeps the
    //         loads "pending" into the next iteration and parks a vmcnt(0) in front of pass 1); then dQ out ----
    consume(kf2, vf2, lse2v, of2);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf[kt][ks] = kf2[kt][ks]; vf[kt][ks] = vf2[kt][ks]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) of[i] = of2[i];
    lse_v = lse2v;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qr = 32 * w + 16 * qt + l15;
      if (qr < a.T && !(a.dbg & 8)) {
        bf16* dst = a.dqkv + ((size_t)b * a.T + qr) * ld + h * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(dst + 16 * dt) = pack4(dq[dt][qt]);
      }
    }
    if (!more) break;
    // ---- 9. every wave is done with the K / V tiles: refill them for the next item ----
    barrier_lds();
    item = next; base = nbase; gbase = ngbase; obase = nobase; b = nb; h = nh;
    cur ^= 1;              // (the K / V tiles are refilled from inside the next pass 1)
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

static int g_attn_dbg = 0;
// 0 (default) = the staged kernel, 1 = the persistent LDS-DMA pipeline (bit-identical results).  The pipeline is NOT the default:
// measured at batch 256 it runs 64-68 us against 58-62 us (tools/bench_attn.py, profiles/r03_attn_bwd_ablation.json).  Its
// ablations say why: the two passes alone take 36 us (12 us per (image, head): 6 us of matrix-pipe time on the busiest SIMD
// plus the exponentials and ~1000 non-loop instructions per item), the tile refills add 10 us and the stores 8 us ON TOP of
// them although every refill is issued a pass ahead -- with all 256 workgroups in lockstep the chip sees 13 MB request bursts
// whose delivery (3-5 us) is longer than the pass they were meant to hide behind.  Environment ROVIT_ATTN_BWD_PIPE=1 enables it.
static int g_attn_bwd_pipe = [] { const char* e = getenv("ROVIT_ATTN_BWD_PIPE"); return e ? atoi(e) : 0; }();   // 0 two passes, 1 persistent, 2 ring
extern "C" int rovit_set_attn_bwd_pipe(int on) { g_attn_bwd_pipe = on; return ROVIT_OK; }
extern "C" int rovit_set_attn_debug(int d) { g_attn_dbg = d; return ROVIT_OK; }

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
  a.dout = (const bf16*)dout; a.dqkv = (bf16*)dqkv; a.dbg = g_attn_dbg;
  // round 3 (opt-in, see g_attn_bwd_pipe): persistent workgroups, tiles by LDS-DMA one phase ahead (attn_bwd_pipe_kernel);
  // ROVIT_ATTN_BWD_WGS=n: workgroups of the persistent launch (default: one per CU)
  if (g_attn_bwd_pipe == 2) {
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_ring_kernel, ATTN_RING_LDS), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(attn_bwd_ring_kernel, dim3(batch * heads), dim3(NW * 64), ATTN_RING_LDS, (hipStream_t)stream, a);
    ROVIT_CHECK_LAUNCH("attn_bwd_ring_kernel");
    return ROVIT_OK;
  }
  const bool use_pipe = g_attn_bwd_pipe == 1;
  static const int wgs_env = getenv("ROVIT_ATTN_BWD_WGS") ? atoi(getenv("ROVIT_ATTN_BWD_WGS")) : 256;
  if (use_pipe) {
    const int items = batch * heads;
    const int wgs = items < wgs_env ? items : (wgs_env < 1 ? 1 : wgs_env);
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_pipe_kernel, ATTN_PIPE_LDS), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(attn_bwd_pipe_kernel, dim3(wgs), dim3(NW * 64), ATTN_PIPE_LDS, (hipStream_t)stream, a, items);
    ROVIT_CHECK_LAUNCH("attn_bwd_pipe_kernel");
    return ROVIT_OK;
  }
  const size_t lds = (size_t)4 * TP * AST * sizeof(bf16) + 2 * TP * sizeof(float);
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)attn_bwd_kernel, lds), ROVIT_ERR_LAUNCH, "attention_bwd: cannot raise the LDS limit");
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(batch * heads), dim3(NW * 64), lds, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("attn_bwd_kernel");
  return ROVIT_OK;
}
