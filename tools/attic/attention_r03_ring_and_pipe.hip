// Round-3 experiments removed from the product in round 4: attn_bwd_ring_kernel (one pass, dQ in a rotating LDS tile:
// 71 us against 57) and attn_bwd_pipe_kernel (persistent, LDS-DMA prefetch: 64-68 us).  Kept for reference only; not compiled.
// They were written against csrc/attention.hip at commit 4feb417 (helpers stage_tile, row_frag, tr_issue, ... live there).
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


