// The MLP half of a DeiT-Tiny block in ONE launch (round 3):
//
//     X[m,:] += fc2( GELU( fc1( xhat2[m,:] ) ) ),   then the LayerNorm that follows the residual add (xhat_out, rstd_out)
//
// Reference arithmetic being restated: timm `Mlp` (fc1 192 -> 768, exact-erf GELU, fc2 768 -> 192) inside a pre-norm block and
// the next block's norm1, reached through /root/reference/models/backbone.py:23-25.  Before this kernel
// the half was two launches (fc1 + GELU writing `act` / `gelu'`, then fc2 + residual + LayerNorm re-reading `act`): 349 MB and
// 107 us per block at batch 256; `act` (77.5 MB per block) is now never read back in the forward, and an inference call
// writes neither `act` nor `gelu'`.
//
// Structure.  A workgroup (8 waves) owns 256 rows; wave w owns the two 16-row tiles w and w + 8 for the WHOLE chain, so no
// activation ever crosses waves:
//   * xhat2 fragments of the wave's 32 rows stay in registers (MFMA B operand, 48 registers);
//   * the hidden dimension is walked in 24 chunks of 32 units.  Per chunk the wave computes pre^T[hidden][row] with the
//     weights as the MFMA "A" operand (v_mfma_f32_16x16x32_bf16), applies GELU in registers, and the bf16 result IS the B
//     operand of the fc2 product over that chunk: the rows of W1 inside a 16-row MFMA tile are PERMUTED by the weight
//     preparation (tile a holds hidden units 8q + r, tile b 8q + 4 + r for accumulator row 4q + r), so that lane (row, q)
//     ends up with the eight consecutive hidden units 8q .. 8q+7 of its row -- one 16-byte store each for `act` and
//     `gelu'`, and the natural contraction order for fc2;
//   * fc2 accumulates out^T[192][row] over the 24 chunks in 96 accumulator registers;
//   * both weight matrices (590 KB of bf16: neither registers nor LDS hold them next to the tiles) are STREAMED per workgroup
//     from L2 through a 3-slot LDS ring by LDS-DMA (global_load_lds_dwordx4): the preparation (rovit_mlp_prepare_stream)
//     writes them as the exact LDS image, fragment-major -- chunk c = 24 pieces of 1 KB, piece = the 64 lanes' 16-byte MFMA
//     A-fragments -- so the DMA is a linear copy and every fragment read is lane-linear (conflict-free ds_read_b128).  All
//     workgroups walk the same stream in the same order: after the first reader of an XCD it is served by that XCD's L2;
//   * completion is hand-counted: per chunk a wave issues 3 DMA pieces and S stores (buffer stores, so that rows beyond M
//     are dropped by the bounds check and the COUNT is the same in every wave), vmcnt(3 + 2 S) leaves two chunks in flight;
//     one barrier per chunk;
//   * epilogue: bf16(out + bias) is staged in LDS (aliasing the ring) and a row-wise pass (16 lanes per row) adds it to the
//     fp32 residual stream and computes the next LayerNorm, exactly the arithmetic of gemm_ws_kernel's EPI_RESID_LN epilogue.
// BACKWARD (round 3, same skeleton, KIND = 1): the dgrad chain of the MLP half in one launch,
//     dpre = (dY W2T^T) * gelu'(pre)   [kept: the fc1 weight gradient needs it]      dxhat2 = dpre W1T^T
//     dX += rstd2 (dxhat2 - mean(dxhat2) - xhat2 mean(dxhat2 xhat2)),  dXb = bf16(dX)        (LayerNorm-2 backward)
// i.e. rovit_gemm_nt(EPI_MUL) + rovit_gemm_ln_bwd without re-reading dpre (77.5 MB per block at batch 256).  The stream is the
// same image built from (W2T, W1T) instead of (W1, W2); the gelu' tile of a chunk (each wave's own two 16 x 32 pieces) comes
// in through the SAME ring by LDS-DMA with a per-lane source address (a register load inside the loop would make the compiler
// wait vmcnt(0) and drain the ring), and `x gelu'` replaces GELU.  dpre is bit-identical to the two-launch path.
// fc1 sums in the same order as gemm_ws_dma_kernel<EPI_GELU> (same MFMA, same k order, bias as the initial accumulator), so
// `act` and `gelu'` are BIT-IDENTICAL to the two-launch path; fc2 sums the hidden units in one chain instead of two halves
// (the K = 768 kernel splits K over wave pairs), so X agrees to fp32 summation order (tests/test_gpu_round3.py).
#include <cstdlib>
#include "common.h"

// cache policy of the kept-activation stores (act, gelu'): 0 default, 2 = nt (written once, read by the backward 100+ MB of traffic later)
#ifndef ROVIT_ACT_STORE_AUX
#define ROVIT_ACT_STORE_AUX 0
#endif
#ifndef ROVIT_MLP_PIPE_HINTS
#define ROVIT_MLP_PIPE_HINTS 0
#endif
#ifndef ROVIT_MLP_PIPE_SKEW
#define ROVIT_MLP_PIPE_SKEW 1
#endif

#ifdef ROVIT_DEV
#define MLP_DBG(g, bit) ((g).dbg & (bit))
#else
#define MLP_DBG(g, bit) (false)    // the product kernels have no skip-work path
#endif

namespace {

constexpr int D = 192, HID = 768, HC = 32, NCHUNK = HID / HC;
constexpr int PIECE = 512;                    // bf16 elements of a 1 KB piece (64 lanes x 16 bytes)
constexpr int CH_PIECES = 24;                 // 12 first-GEMM fragments (2 tiles x 6 k-steps) + 12 second-GEMM fragments (12 output tiles)
constexpr int CH_ELEMS = CH_PIECES * PIECE;   // 24 KB of weights per chunk
constexpr int NSLOT = 3;
#ifdef ROVIT_DEV
constexpr int TAIL_WAVES_DEFAULT = 8;     // waves per workgroup of the forward block tail (16: one row tile per wave)
#endif
// pipelined forward: ring entry j carries the fc2 fragments of hidden chunk j - PSKEW next to the fc1 fragments of chunk j.
// PSKEW = 2: iteration j issues fc1 of chunk j, the GELU look-ups of chunk j - 1 and fc2 of chunk j - 2 -- three mutually
// independent streams of work (matrix, vector + LDS gather, matrix); 1: fc2 of chunk j - 1 behind its own GELU.
constexpr int PSKEW = ROVIT_MLP_PIPE_SKEW;
constexpr int PENTRIES = NCHUNK + PSKEW;
// "block tail" forward (TAIL): the attention-output projection, its residual add and norm2 run IN FRONT of the MLP chain in the same
// launch -- three more ring entries (the 192 x 192 proj weight as 12 output tiles x 6 k-steps) ahead of the skewed MLP image.  The
// residual stream then never leaves the registers between the two halves: X is read once and written once per block (155 MB less
// per block than proj + LayerNorm and the MLP half as two launches) and xhat2 is not read back.  Output tile ot of the proj product
// (and, by the same permutation of its weight rows, of fc2) holds the columns tail_col(ot, row) on its accumulator rows, so that the
// tile pair (2k, 2k+1) gives a lane EIGHT consecutive columns 32k + 8lg .. +7 of its token row: after norm2 those are, as they
// stand, the fc1 B fragment of k-step k, and X / xhat leave as 32- / 16-byte runs.
constexpr int TPROJ = 3;                      // ring entries of the proj weight
constexpr int TQKV = 9;                       // ring entries of the NEXT block's qkv weight (36 output tiles x 6 k-steps), behind the MLP image
constexpr int TENTRIES = TPROJ + PENTRIES + TQKV;
__host__ __device__ constexpr int tail_col(int ot, int row) { return 32 * (ot >> 1) + 8 * (row >> 2) + 4 * (ot & 1) + (row & 3); }
constexpr int CSTR = 192 + 8;                 // staged output tile [ROWS][CSTR] bf16
// NW waves per workgroup, 32 rows per wave.  NW = 8: one 256-row workgroup per CU.  NW = 4 (forward only): 128-row workgroups,
// two per CU (76 KB each).  A backward slot also holds the workgroup's gelu' tile of the chunk (2 NW pieces).
constexpr int slot_elems(int kind, int nw) { return CH_ELEMS + (kind ? 2 * nw * PIECE : 0); }
constexpr int region_elems(int kind, int nw) {      // ring / staged tile (aliased)
  return NSLOT * slot_elems(kind, nw) > 32 * nw * CSTR ? NSLOT * slot_elems(kind, nw) : 32 * nw * CSTR;
}
// GELU table of the pipelined forward.  The GELU input is the bf16-rounded pre-activation, so gelu / gelu' are functions of a 16-bit
// pattern: for 2^-24 <= |x| < 16 (28 exponents x 128 mantissas x 2 signs) the pair (bf16 gelu, bf16 gelu') is looked up in a 32 KB
// LDS table that the weight preparation fills WITH gelu_and_grad itself (bit-identical by construction); any other input (zero,
// denormal-small, huge, NaN) takes the formula under a wave-uniform branch.  Entry of pattern u: index ((u & 0x7FFF) - GT_EM_LO),
// negative inputs 4096 entries further.
constexpr int GT_EM_LO = (127 - 24) << 7, GT_N = 28 << 7, GT_NEG = 4096, GT_ENTRIES = 8192;
constexpr size_t lds_bytes(int kind, int nw) { return (size_t)region_elems(kind, nw) * sizeof(bf16) + (HID + D) * sizeof(float); }

struct MlpArgs {
  const bf16* xin;        // (M,192): forward xhat2, backward dY (gradient w.r.t. the MLP output)
  const bf16* wstream;    // rovit_mlp_prepare_stream image: forward (W1 folded, W2), backward (W2T, W1T folded)
  const float* b1;        // forward: (768) folded fc1 bias
  const float* b2;        // forward: (192) fc2 bias
  bf16* act;              // forward gelu(pre) (MODE >= 1); backward dpre (output).  CHUNK-MAJOR [24][rows][32], see the kernel
  bf16* dact;             // forward gelu'(pre) (MODE == 2), chunk-major
  const bf16* mul;        // backward gelu'(pre), chunk-major [24][M][32]
  int act_rows;           // forward: rows of the whole chunk-major tensors (>= M: a launch may cover a row range of them)
  int rpw;                // token rows a workgroup takes (<= 32 NW; fewer = more workgroups, the surplus row tiles idle)
  const float* bp;        // TAIL: (192) proj bias
  bf16* dO;               // backward TAIL: (M,192) gradient w.r.t. the attention output (= dXb Wproj), or NULL
  const float* bq;        // forward TAIL: (576) folded qkv bias of the NEXT block
  bf16* qkv;              // forward TAIL: (M,576) the next block's qkv projection of xhat_out, or NULL (no qkv phase)
  bf16* xhat2;            // TAIL: norm2 output (M,192) kept for the backward (NULL: inference)
  float* rstd2;           // TAIL: (M)
  const bf16* gelu_table; // pipelined forward: the 32 KB table behind the two stream images
  float* X;               // (M,192) forward: residual stream; backward: dX; updated in place
  bf16* xhat;             // forward: next LayerNorm output or NULL; backward: xhat2 (input)
  float* rstd;            // forward: (M) output; backward: rstd2 (input)
  bf16* xb;               // backward: (M,192) bf16 copy of the updated dX
  float eps;
  int M;
#ifdef ROVIT_DEV
  int dbg;                // developer library only (ROVIT_KNOB_MLP_DBG, timing ablations): bit 0 skip the row-wise epilogue, 1 skip GELU, 2 skip fc2, 3 skip fc1
#endif
};

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KIND 0 forward (MODE 0: inference, nothing kept; 1: keep act; 2: keep act and gelu'), KIND 1 backward (MODE 1: dpre kept)
// STAG (forward, 8 waves): waves 4-7 run HALF A CHUNK behind waves 0-3 in program order (MI355X guide, "two waves per SIMD",
// item 9).  The two waves of a SIMD otherwise move in lockstep through [fc1: matrix pipe] [GELU: vector pipe] [fc2: matrix pipe]:
// they contend for the matrix pipe, then both sit in the ~300-instruction GELU that a single wave issues at half the vector
// rate, and nothing overlaps (measured: 3400 cycles per chunk for 1536 cycles of matrix time).  Staggered,
//   waves 0-3:  fc1(c)          | GELU tile 0, tile 1 (c)         | fc2(c)
//   waves 4-7:  GELU tile 1 (c-1) | fc2(c-1) | fc1(c)              | GELU tile 0 (c)
// one wave's matrix segments fall into its partner's GELU.  Waves 4-7 carry the tile-1 pre-activations and the tile-0 GELU
// output of chunk c-1 across the barrier (12 registers), read chunk c-1's fc2 fragments one iteration later (so the ring has
// FOUR slots: 96 KB, still under the staged output tile it aliases), issue out-of-range (dropped, but counted) stores in
// their first iteration so that the vmcnt arithmetic is the same for every wave, and finish chunk 23 behind the loop.
// Outputs are bit-identical to the unstaggered kernel (same operations on the same values, per wave in the same order).
// PIPE (forward, 8 waves): software pipeline INSIDE each wave.  The weight stream is the SKEWED image (ring entry j = fc1 fragments of
// hidden chunk j | fc2 fragments of hidden chunk j - PSKEW, PENTRIES entries), so iteration j issues the fc1 MFMAs of chunk j,
// the GELU of chunk j - 1 (whose pre-activations were packed to bf16 at the end of iteration j - 1) and the fc2 MFMAs of chunk
// j - 2 (whose GELU output iteration j - 1 left in registers): nothing inside an iteration depends on anything else in it, so
// the scheduler can put the vector work and the LDS gathers into the matrix pipe's shadow.  The GELU itself is a table look-up
// (GT_*).  The first iterations' GELU / fc2 run on zeros and the last ones' fc1 on zero blocks; stores of chunks that do not
// exist are sent out of range (dropped, but counted).
// TPW (tiles per wave) = 1, NW = 16 (round 4, forward block tail only): SIXTEEN waves of ONE 16-row tile each -- four waves per SIMD
// instead of two.  With two waves per SIMD the kernel runs as the SUM of its matrix, vector and LDS time (34 + 30 + 33 % of the
// launch, profiles/r04_pmc_sq.json): both waves of a SIMD sit in the same phase of the same iteration.  Four waves (<= 128 registers:
// the fc2 accumulators of one tile are 48, not 96) give the scheduler matrix work of one wave for the vector / LDS work of another.
// Price: a weight fragment read from LDS serves one row tile instead of two (LDS reads x 2).  MEASURED (tools/ab_tail.sh, one box,
// M = 50 432): 97.5 against 92.6 us for the training launch, 87.9 against 84.0 for inference -- bit-identical outputs, SLOWER: the
// doubled fragment reads and the 16-wave barrier cost more than the occupancy gives.  Kept in the developer library (knob 18) only.
template <int KIND, int MODE, int NW, bool STAG = false, bool PIPE = false, bool TAIL = false, int TPW = 2>
__global__ __launch_bounds__(NW * 64, TPW == 1 ? 4 : 2) void mlp_fused_kernel(const MlpArgs g) {
  static_assert(TPW == 2 || (TPW == 1 && PIPE && TAIL && KIND == 0 && NW == 16), "one tile per wave: the 16-wave forward block tail");
  static_assert(!TAIL || PIPE || KIND == 1, "the forward block tail builds on the pipelined forward");
  constexpr int NBIAS = HID + D + (TAIL ? D + 3 * D : 0);       // floats behind the ring: b1, b2 [, proj bias, next block's qkv bias]
  static_assert(!STAG || (KIND == 0 && NW == 8), "the staggered schedule is the 8-wave forward's");
  static_assert(!PIPE || (KIND == 0 && (NW == 8 || NW == 16) && !STAG), "the pipelined schedule is the 8- / 16-wave forward's");
  constexpr int S = TPW * MODE;               // stores one wave issues per chunk
  // (rows per workgroup = 16 * TPW * NW: wave w owns the 16-row tiles w and w + NW)
  // DMA pieces one wave issues per chunk: 24 / NW, or (NW = 16) two for waves 0-7 and one for waves 8-15 -- the counted waits differ by wave
  constexpr int PWHI = (CH_PIECES + NW - 1) / NW, PWLO = CH_PIECES / NW, PWREM = CH_PIECES % NW;
  constexpr int PW = PWLO + (KIND ? 2 : 0);
#define WAIT_DMA(extra) do { if (PWHI == PWLO || w >= PWREM) wait_vm<PWLO + (KIND ? 2 : 0) + (extra)>(); else wait_vm<PWHI + (KIND ? 2 : 0) + (extra)>(); } while (0)
  constexpr int SLOT = slot_elems(KIND, NW);
  constexpr int J0 = (TAIL && !KIND) ? TPROJ : 0;                  // forward block tail: ring entries in front of the MLP image
  const int NTOT = PENTRIES + J0 + ((TAIL && !KIND && g.qkv) ? TQKV : 0);   // ... and in all (wave-uniform)
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];   // ONE array: ring / staged tile, then the two biases
  // the ring (aliased by the generic epilogue's staged output tile; the one-tile-per-wave block tail stages nothing)
  constexpr int REGION = TPW == 1 ? NSLOT * slot_elems(KIND, NW) : region_elems(KIND, NW);
  float* s_bias = (float*)(lds + REGION);       // forward: [768] b1, [192] b2
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int r0 = blockIdx.x * g.rpw;

  // this wave's rows: tile i holds rows r0 + 16 (w + NW i) + l15 (clamped for the loads; stores are bounds-checked)
  int mrow[TPW], mcl[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int local = 16 * NW * i + 16 * w + l15;
    mrow[i] = local < g.rpw ? r0 + local : 0x3FFFFFFF;      // rows this workgroup does not own count as beyond M
    mcl[i] = mrow[i] < g.M ? mrow[i] : g.M - 1;
  }
  auto dma = [&](int chunk, int slot) {
    const bf16* src = g.wstream + (size_t)chunk * CH_ELEMS + lane * 8;
#pragma unroll
    for (int q = 0; q < PWHI; ++q) {
      const int piece = w + NW * q;
      if (PWHI == PWLO || piece < CH_PIECES)          // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * PIECE),
                                         (__attribute__((address_space(3))) void*)(lds + slot * SLOT + piece * PIECE), 16, 0, 0);
    }
    if (KIND) {        // this wave's own gelu' pieces: lane (row l15, q = lg) <- gelu'[row][32 chunk + 8 q .. +7]
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.mul + ((size_t)chunk * g.M + mcl[i]) * HC + lg * 8),
                                         (__attribute__((address_space(3))) void*)(lds + slot * SLOT + (CH_PIECES + w + NW * i) * PIECE), 16, 0, 0);
    }
  };
  if (!KIND && tid < NBIAS / 4) {                 // biases -> LDS (240 / 288 float4)
    float4 v;
    if (tid < HID / 4) v = ((const float4*)g.b1)[tid];
    else if (tid < (HID + D) / 4) v = ((const float4*)g.b2)[tid - HID / 4];
    else if (tid < (HID + 2 * D) / 4) v = ((const float4*)g.bp)[tid - (HID + D) / 4];
    else v = g.bq ? ((const float4*)g.bq)[tid - (HID + 2 * D) / 4] : make_float4(0.f, 0.f, 0.f, 0.f);
    ((float4*)s_bias)[tid] = v;
  }
  bf16x8 xf[TPW][6];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) xf[i][ks] = *(const bf16x8*)(g.xin + (size_t)mcl[i] * D + ks * 32 + lg * 8);
  // buffer resources for the kept activations: rows >= M fall outside num_records and are dropped by the hardware
  __amdgpu_buffer_rsrc_t r_act, r_dact;
  // The kept activations are CHUNK-MAJOR, [24][rows][32] (32 hidden units = 64 bytes per row and chunk): the store instruction of a
  // 16-row tile then writes ONE contiguous kilobyte instead of 16 pieces of 64 bytes 1 536 bytes apart.  Row-major, the launch
  // took 94 us once the outputs no longer fit the 256 MB Infinity Cache (as in the training step) against 71 us in a loop over
  // one buffer set; chunk-major 76 against 69 us (profiles/r03_mlp_store_layout.json: a timing hack that preceded this layout).
  // cblk = bytes from one chunk to the next; rows beyond M get an offset that stays outside num_records for every chunk.
  const unsigned cblk = (unsigned)(KIND ? g.M : g.act_rows) * (HC * 2);
  const int nrec = (int)((size_t)(NCHUNK - 1) * cblk + (size_t)g.M * (HC * 2));
  __amdgpu_buffer_rsrc_t r_qkv;
  if (TAIL && !KIND) r_qkv = __builtin_amdgcn_make_buffer_rsrc((void*)g.qkv, 0, g.qkv ? (int)((size_t)g.M * 3 * D * 2) : 0, 0x00020000);
  if (MODE >= 1) r_act = __builtin_amdgcn_make_buffer_rsrc((void*)g.act, 0, nrec, 0x00020000);
  if (MODE == 2) r_dact = __builtin_amdgcn_make_buffer_rsrc((void*)g.dact, 0, nrec, 0x00020000);
  unsigned soff[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) soff[i] = mrow[i] < g.M ? (unsigned)mrow[i] * (HC * 2) + lg * 16 : 0xF0000000u;

  f32x4 a2[12][TPW];
#pragma unroll
  for (int ot = 0; ot < 12; ++ot)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if constexpr (TAIL && KIND == 0) {        // the residual stream itself, in the (permuted) accumulator layout: everything is accumulated onto it
        const float4 x = *(const float4*)(g.X + (size_t)mcl[i] * D + tail_col(ot, 4 * lg));
        a2[ot][i] = (f32x4){x.x, x.y, x.z, x.w};
      } else if constexpr (TAIL && KIND == 1) {  // backward: dX itself (scaled below), so that nothing but xhat2 is loaded behind the loop
        const float4 x = *(const float4*)(g.X + (size_t)mcl[i] * D + tail_col(ot, 4 * lg));
        a2[ot][i] = (f32x4){x.x, x.y, x.z, x.w};
      } else {
        a2[ot][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }

  const bf16* gtab = lds + REGION + 2 * NBIAS;       // PIPE: the GELU table (behind the biases)
  if constexpr (PIPE) {
    const bf16* src = g.gelu_table + lane * 8;
#pragma unroll
    for (int r = 0; r < 32 / NW; ++r)                                    // 32 pieces of 1 KB: 8 waves x 4 / 16 x 2 (older than the ring's first DMA)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (r * NW + w) * PIECE),
                                       (__attribute__((address_space(3))) void*)(lds + REGION + 2 * NBIAS + (r * NW + w) * PIECE), 16, 0, 0);
  }
  // (the ring's first two entries are requested BEHIND the prologue's register loads: a wait for those then leaves the DMA in flight)
  dma(0, 0);
  dma(1, 1);

  // Backward block tail: the norm2 backward dX += rstd (g - mean(g) - h mean(g h)) is LINEAR in g = dxhat2, so the accumulators start
  // as A0 = dX / rstd and the fc1-dgrad chain adds g on top; behind the loop dX_out = rstd (A - c1 - h c2) with
  // c1 = mean(A) - mean(dX) / rstd, c2 = mean(A h) - mean(dX h) / rstd: dX is read HERE, at the top of the launch, not behind the loop.
  float bm1[2] = {0.f, 0.f}, bm2[2] = {0.f, 0.f}, binv[2] = {1.f, 1.f};
  if constexpr (TAIL && KIND == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t mo = (size_t)mcl[i] * D;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const bf16x8 h = *(const bf16x8*)(g.xhat + mo + 32 * k + 8 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += a2[2 * k][i][r] + a2[2 * k + 1][i][r];
          s2 += a2[2 * k][i][r] * (float)h[r] + a2[2 * k + 1][i][r] * (float)h[4 + r];
        }
      }
      binv[i] = 1.f / g.rstd[mcl[i]];
      bm1[i] = group4_sum(s1) * (1.f / 192.f) * binv[i];
      bm2[i] = group4_sum(s2) * (1.f / 192.f) * binv[i];
#pragma unroll
      for (int ot = 0; ot < 12; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) a2[ot][i][r] *= binv[i];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // bias ds_writes retired before the first barrier
  if constexpr (STAG) {
    static_assert(4 * CH_ELEMS <= region_elems(0, 8), "four ring slots must fit under the staged output tile");
    const bool late = w >= NW / 2;                            // wave-uniform
    // The xhat2 fragments of tile 1 live in LDS (48 KB behind the biases: this wave's own 6 KB, lane-linear, written once): with
    // both tiles' fragments in registers the staggered loop needs ~265 registers and the allocator spills two fragments, whose
    // reloads inside the loop are vector-memory operations that drain the ring
    bf16* xl = lds + region_elems(0, NW) + 2 * (HID + D) + w * 6 * PIECE + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) *(bf16x8*)(xl + ks * PIECE) = xf[1][ks];
    auto fc1 = [&](const bf16* sb, int c, f32x4 (&a1)[2][2]) {
      const f32x4 ba = *(const f32x4*)(s_bias + c * HC + 8 * lg), bb = *(const f32x4*)(s_bias + c * HC + 8 * lg + 4);
      a1[0][0] = ba; a1[0][1] = ba; a1[1][0] = bb; a1[1][1] = bb;
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const bf16x8 wa = *(const bf16x8*)(sb + ks * PIECE), wb = *(const bf16x8*)(sb + (6 + ks) * PIECE);
        const bf16x8 x1 = *(const bf16x8*)(xl + ks * PIECE);
        if MLP_DBG(g, 8) continue;
        a1[0][0] = mfma16(wa, xf[0][ks], a1[0][0]);
        a1[1][0] = mfma16(wb, xf[0][ks], a1[1][0]);
        a1[0][1] = mfma16(wa, x1, a1[0][1]);
        a1[1][1] = mfma16(wb, x1, a1[1][1]);
      }
    };
    auto gelu_tile = [&](const f32x4& pa, const f32x4& pb, bf16x8& av, bf16x8& dv) {      // tile a / tile b rows of ONE 16-row tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ga, gd;
        if MLP_DBG(g, 2) { av[r] = (bf16)pa[r]; dv[r] = av[r]; av[4 + r] = (bf16)pb[r]; dv[4 + r] = av[4 + r]; continue; }
        gelu_and_grad((float)(bf16)pa[r], ga, gd);
        av[r] = (bf16)ga; dv[r] = (bf16)gd;
        gelu_and_grad((float)(bf16)pb[r], ga, gd);
        av[4 + r] = (bf16)ga; dv[4 + r] = (bf16)gd;
      }
    };
    auto store_tile = [&](int i, int c, const bf16x8& av, const bf16x8& dv) {
      if (MODE >= 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, av), r_act, soff[i] + c * cblk, 0, 0);
      if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, dv), r_dact, soff[i] + c * cblk, 0, 0);
    };
    auto fc2 = [&](const bf16* sb, const bf16x8& av0, const bf16x8& av1) {
#pragma unroll
      for (int ot = 0; ot < 12; ++ot) {
        const bf16x8 w2 = *(const bf16x8*)(sb + (12 + ot) * PIECE);
        if MLP_DBG(g, 4) continue;
        a2[ot][0] = mfma16(w2, av0, a2[ot][0]);
        a2[ot][1] = mfma16(w2, av1, a2[ot][1]);
      }
    };
    // Half-steps of a chunk c:  H2(c) = fc1 of chunk c (matrix pipe), GELU of its tile 0 (vector pipe), stores of tile 0;
    //                           H1(c+1) = GELU of tile 1 of chunk c, its stores, fc2 of chunk c (both tiles).
    // Waves 0-3 run H2(c) | H1(c+1) in iteration c (the lockstep order with a barrier in the middle), waves 4-7 run H1(c) | H2(c):
    // in every half-iteration one wave of a SIMD is in [matrix, vector] order and its partner in [vector, matrix] order.  The two
    // groups have their own straight-line loops (one loop with a role branch inside made hipcc shuttle the 96 accumulator
    // registers through ~150 v_mov_b64 per half-step).  Chunk c is read from the top of iteration c until the first half of
    // iteration c + 1 (waves 4-7's fc2), its DMA is issued at the top of iteration c - 2 into ring slot c % 4, whose previous
    // tenant, chunk c - 4, was last read in iteration c - 3.
    bf16x8 keep_pre = {};            // bf16(pre-activation) of tile 1: elements 0-3 tile a rows, 4-7 tile b rows
    bf16x8 keep_av0 = {};
    auto top = [&](int c) {
      if (c == 0) wait_vm<PW>();
      else if (c == 1) wait_vm<PW + S>();
      else if (c < NCHUNK - 1) wait_vm<PW + 2 * S>();
      else wait_vm<2 * S>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (c + 2 < NCHUNK) dma(c + 2, (c + 2) & 3);
    };
    auto mid = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    auto H2 = [&](int c) {
      const bf16* sb = lds + (c & 3) * SLOT + lane * 8;
      f32x4 a1[2][2];
      fc1(sb, c, a1);
#pragma unroll
      for (int r = 0; r < 4; ++r) { keep_pre[r] = (bf16)a1[0][1][r]; keep_pre[4 + r] = (bf16)a1[1][1][r]; }
      asm volatile("" : "+v"(keep_pre));                    // packed NOW: the eight fp32 registers of tile 1 are free during the GELU
      bf16x8 dv0;
      gelu_tile(a1[0][0], a1[1][0], keep_av0, dv0);
      store_tile(0, c, keep_av0, dv0);
    };
    auto H1 = [&](int c) {                                   // second half of chunk c - 1
      bf16x8 av1, dv1;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float ga, gd;
        gelu_and_grad((float)keep_pre[r], ga, gd);
        av1[r] = (bf16)ga; dv1[r] = (bf16)gd;
      }
      store_tile(1, c - 1, av1, dv1);
      fc2(lds + ((c - 1) & 3) * SLOT + lane * 8, keep_av0, av1);
    };
    if (!late) {
#pragma unroll 1
      for (int c = 0; c < NCHUNK; ++c) {
        top(c);
        H2(c);
        mid();
        H1(c + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      mid();                                                 // pairs with the barrier in front of the other group's last half-step
    } else {
      top(0);
      {                                                      // nothing to finish yet: the same store count, out of range (dropped, but counted)
        const u32x4v z = {0u, 0u, 0u, 0u};
        if (MODE >= 1) __builtin_amdgcn_raw_buffer_store_b128(z, r_act, 0xFFFFFF00u, 0, 0);
        if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(z, r_dact, 0xFFFFFF00u, 0, 0);
      }
      mid();
      H2(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
      for (int c = 1; c < NCHUNK; ++c) {
        top(c);
        H1(c);
        mid();
        H2(c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      mid();
      H1(NCHUNK);                                            // chunk 23's second half
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else if constexpr (PIPE) {
    bf16x8 pk[TPW] = {};             // bf16(pre-activation) of chunk j - 1: tile i, elements 4 t + r
    bf16x8 avo[TPW] = {};            // PSKEW = 2: gelu of chunk j - 2
    int slot = 0;

    // every iteration issues [DMA(J+2): PW] [S stores] (iterations without real stores send theirs out of range: dropped, but
    // counted), so the counted waits are those of the lockstep loop with NTOT entries
    auto top = [&](int J) {
      if (J == 0) WAIT_DMA(0);
      else if (J == 1) WAIT_DMA(S);
      else if (J < NTOT - 1) WAIT_DMA(2 * S);
      else wait_vm<2 * S>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (J + 2 < NTOT) dma(J + 2, slot == 0 ? 2 : slot - 1);
    };
    if constexpr (TAIL) {
      // ---- the attention-output projection onto the residual stream: xf holds the attention output's fragments, a2 holds X ----
#pragma unroll
      for (int J = 0; J < TPROJ; ++J) {
        top(J);
        const bf16* sb = lds + slot * SLOT + lane * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 wp = *(const bf16x8*)(sb + (6 * u + ks) * PIECE);
#pragma unroll
            for (int i = 0; i < TPW; ++i) a2[4 * J + u][i] = mfma16(wp, xf[i][ks], a2[4 * J + u][i]);
          }
        {
          const u32x4v z = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            if (MODE >= 1) __builtin_amdgcn_raw_buffer_store_b128(z, r_act, 0xFFFFFF00u, 0, 0);
            if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(z, r_dact, 0xFFFFFF00u, 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
      }
      // ---- + proj bias; norm2 in registers; xhat2 = the fc1 B fragments; then + fc2 bias (fc2 accumulates on top) ----
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        float sum = 0.f;
#pragma unroll
        for (int ot = 0; ot < 12; ++ot) {
          const f32x4 b = *(const f32x4*)(s_bias + HID + D + tail_col(ot, 4 * lg));
#pragma unroll
          for (int r = 0; r < 4; ++r) { a2[ot][i][r] += b[r]; sum += a2[ot][i][r]; }
        }
        const float mean = group4_sum(sum) * (1.f / 192.f);
        float qs = 0.f;
#pragma unroll
        for (int ot = 0; ot < 12; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = a2[ot][i][r] - mean; qs += v * v; }
        const float rs = rsqrtf(group4_sum(qs) * (1.f / 192.f) + g.eps);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          f32x4 lo, hi;
#pragma unroll
          for (int r = 0; r < 4; ++r) { lo[r] = (a2[2 * k][i][r] - mean) * rs; hi[r] = (a2[2 * k + 1][i][r] - mean) * rs; }
          xf[i][k] = pack8(lo, hi);
        }
        if (g.xhat2 && mrow[i] < g.M) {
#pragma unroll
          for (int k = 0; k < 6; ++k) *(bf16x8*)(g.xhat2 + (size_t)mrow[i] * D + 32 * k + 8 * lg) = xf[i][k];
          if (lg == 0) g.rstd2[mrow[i]] = rs;
        }
#pragma unroll
        for (int ot = 0; ot < 12; ++ot) {
          const f32x4 b = *(const f32x4*)(s_bias + HID + tail_col(ot, 4 * lg));
#pragma unroll
          for (int r = 0; r < 4; ++r) a2[ot][i][r] += b[r];
        }
      }
    }
#pragma unroll 1
    for (int j = 0; j < PENTRIES; ++j) {
      top(j + J0);
      const bf16* sb = lds + slot * SLOT + lane * 8;
      const int cb = j < NCHUNK ? j : NCHUNK - 1;             // bias row of a real chunk (the products of the last PSKEW iterations are discarded)
      f32x4 a1[2][TPW];
      {
        const f32x4 ba = *(const f32x4*)(s_bias + cb * HC + 8 * lg), bb = *(const f32x4*)(s_bias + cb * HC + 8 * lg + 4);
#pragma unroll
        for (int i = 0; i < TPW; ++i) { a1[0][i] = ba; a1[1][i] = bb; }
      }
      // ---- fc1 of chunk j (matrix pipe) ... ----
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const bf16x8 wa = *(const bf16x8*)(sb + ks * PIECE), wb = *(const bf16x8*)(sb + (6 + ks) * PIECE);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          a1[0][i] = mfma16(wa, xf[i][ks], a1[0][i]);
          a1[1][i] = mfma16(wb, xf[i][ks], a1[1][i]);
        }
      }
      // ---- ... with the GELU of chunk j - 1 in its shadow: table look-ups (a handful of integer operations and one 4-byte LDS
      // gather per element instead of ~20 fp32 operations with two transcendentals: the lockstep kernel is VALU-bound on those) ----
      bf16x8 av[TPW], dv[TPW];
      {
        unsigned t[TPW][8], ix[TPW][8];
        unsigned mx = 0;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const u32x4v pw = __builtin_bit_cast(u32x4v, pk[i]);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned w32 = pw[d];
            ix[i][2 * d] = (w32 & 0x7FFFu) - (unsigned)GT_EM_LO;
            ix[i][2 * d + 1] = ((w32 >> 16) & 0x7FFFu) - (unsigned)GT_EM_LO;
            mx = max(mx, max(ix[i][2 * d], ix[i][2 * d + 1]));
            const unsigned a0 = (min(ix[i][2 * d], (unsigned)(GT_N - 1)) << 1) | ((w32 & 0x8000u) >> 2);        // in bf16 elements
            const unsigned a1 = (min(ix[i][2 * d + 1], (unsigned)(GT_N - 1)) << 1) | ((w32 >> 18) & 0x2000u);
            t[i][2 * d] = __builtin_bit_cast(unsigned, *(const bf16x2*)(gtab + a0));
            t[i][2 * d + 1] = __builtin_bit_cast(unsigned, *(const bf16x2*)(gtab + a1));
          }
        }
        if (__builtin_amdgcn_ballot_w64(mx >= (unsigned)GT_N)) {          // rare: some lane holds an input outside the table
#pragma unroll
          for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (ix[i][e] >= (unsigned)GT_N) {
                float ga, gd;
                gelu_and_grad((float)pk[i][e], ga, gd);
                const bf16x2 pr = {(bf16)ga, (bf16)gd};
                t[i][e] = __builtin_bit_cast(unsigned, pr);
              }
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          u32x4v a, dd;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            a[d] = __builtin_amdgcn_perm(t[i][2 * d + 1], t[i][2 * d], 0x05040100u);
            dd[d] = __builtin_amdgcn_perm(t[i][2 * d + 1], t[i][2 * d], 0x07060302u);
          }
          av[i] = __builtin_bit_cast(bf16x8, a);
          dv[i] = __builtin_bit_cast(bf16x8, dd);
        }
      }
#if ROVIT_MLP_PIPE_HINTS
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        if (k % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
        __builtin_amdgcn_sched_group_barrier(0x002, ROVIT_MLP_PIPE_HINTS, 0);   // vector instructions
      }
#endif
      if (MODE >= 1) {
        const bool real = j >= 1 && j <= NCHUNK;                // no such chunk: an offset beyond num_records (no wrap-around: absolute)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const unsigned off = real ? soff[i] + (unsigned)(j - 1) * cblk : 0xFFFFFF00u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, av[i]), r_act, off, 0, ROVIT_ACT_STORE_AUX);
          if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, dv[i]), r_dact, off, 0, ROVIT_ACT_STORE_AUX);
        }
      }
      // ---- fc2 of chunk j - PSKEW ----
#pragma unroll
      for (int ot = 0; ot < 12; ++ot) {
        const bf16x8 w2 = *(const bf16x8*)(sb + (12 + ot) * PIECE);
#pragma unroll
        for (int i = 0; i < TPW; ++i) a2[ot][i] = mfma16(w2, PSKEW == 2 ? avo[i] : av[i], a2[ot][i]);
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        avo[i] = av[i];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[i][4 * t + r] = (bf16)a1[t][i][r];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else {
  int slot = 0;
#pragma unroll 1
  for (int c = 0; c < NCHUNK; ++c) {
    // chunk c has landed once at most the younger operations of THIS wave are outstanding.  Issue order per iteration:
    // [DMA(c+2): PW] [stores(c): S]; DMA(0), DMA(1) in the prologue.
    if (c == 0) wait_vm<PW>();
    else if (c == 1) wait_vm<PW + S>();
    else if (c < NCHUNK - 1) wait_vm<PW + 2 * S>();
    else wait_vm<2 * S>();
    __builtin_amdgcn_s_barrier();                             // every wave's pieces of chunk c are in; chunk c-1 fully consumed
    asm volatile("" ::: "memory");
    if (c + 2 < NCHUNK) dma(c + 2, slot == 0 ? 2 : slot - 1);   // (c + 2) % 3: the slot chunk c-1 used

    const bf16* sb = lds + slot * SLOT + lane * 8;
    // ---- first GEMM: pre^T (dgrad: g^T) [32 hidden][32 rows]; forward: bias as the initial accumulator ----
    f32x4 a1[2][2];
    if (!KIND) {
      const f32x4 ba = *(const f32x4*)(s_bias + c * HC + 8 * lg), bb = *(const f32x4*)(s_bias + c * HC + 8 * lg + 4);
      a1[0][0] = ba; a1[0][1] = ba; a1[1][0] = bb; a1[1][1] = bb;
    } else {
      a1[0][0] = a1[0][1] = a1[1][0] = a1[1][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!MLP_DBG(g, 8))
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const bf16x8 wa = *(const bf16x8*)(sb + ks * PIECE), wb = *(const bf16x8*)(sb + (6 + ks) * PIECE);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a1[0][i] = mfma16(wa, xf[i][ks], a1[0][i]);
        a1[1][i] = mfma16(wb, xf[i][ks], a1[1][i]);
      }
    }
    // ---- elementwise in registers: lane (row, q) holds hidden units 32 c + 8 q + {0..3} (tile a) and + {4..7} (tile b) ----
    bf16x8 av[2], dv[2];
    if (!KIND) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float ga, gd;
            if MLP_DBG(g, 2) { ga = a1[t][i][r]; gd = ga; }
            else gelu_and_grad((float)(bf16)a1[t][i][r], ga, gd);      // the two-launch path's GELU sees the bf16-staged pre-activation
            av[i][4 * t + r] = (bf16)ga;
            dv[i][4 * t + r] = (bf16)gd;
          }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 mf = *(const bf16x8*)(sb + (CH_PIECES + w + NW * i) * PIECE);      // this lane's own gelu' values
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)           // the two-launch path multiplies the bf16-staged dgrad by the bf16 mask in fp32
            av[i][4 * t + r] = (bf16)((float)(bf16)a1[t][i][r] * (float)mf[4 * t + r]);
      }
    }
    if (MODE >= 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, av[i]), r_act, soff[i] + c * cblk, 0, 0);
        if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, dv[i]), r_dact, soff[i] + c * cblk, 0, 0);
      }
    }
    // ---- second GEMM: out^T[192][32 rows] += W[:, chunk] act^T ----
    if (!MLP_DBG(g, 4))
#pragma unroll
    for (int ot = 0; ot < 12; ++ot) {
      const bf16x8 w2 = *(const bf16x8*)(sb + (12 + ot) * PIECE);
#pragma unroll
      for (int i = 0; i < 2; ++i) a2[ot][i] = mfma16(w2, av[i], a2[ot][i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's reads of the slot are done before it reaches the next barrier
    slot = slot == 2 ? 0 : slot + 1;
  }

  }

  if constexpr (TAIL && KIND == 1) {
    // ---- backward block tail: a2 = dxhat2^T in the permuted layout (the stream's second-GEMM rows are permuted by tail_col): norm2
    // backward in registers (fp32 dxhat2, nothing staged through bf16), dX / dXb leave as 32- / 16-byte runs, and the bf16 dX tile
    // pairs are, as they stand, the B fragments of the proj dgrad dO = dXb Wproj, whose weight fragments are the image's last three
    // entries (requested only now: the ring is idle, the epilogue's register loads have been consumed) ----
    __builtin_amdgcn_s_barrier();                               // every wave has left the loop: the slots are free
    asm volatile("" ::: "memory");
    if (g.dO) {                                                 // the proj dgrad's weight fragments travel while the norm2 backward runs
#pragma unroll
      for (int e = 0; e < TPROJ; ++e) {
        const bf16* src = g.wstream + (size_t)(NCHUNK + e) * CH_ELEMS + lane * 8;
#pragma unroll
        for (int q = 0; q < CH_PIECES / NW; ++q) {
          const int piece = w + NW * q;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * PIECE),
                                           (__attribute__((address_space(3))) void*)(lds + e * SLOT + piece * PIECE), 16, 0, 0);
        }
      }
    }
    bf16x8 db[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t mo = (size_t)mcl[i] * D;
      bf16x8 hh[6];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        hh[k] = *(const bf16x8*)(g.xhat + mo + 32 * k + 8 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += a2[2 * k][i][r] + a2[2 * k + 1][i][r];
          s2 += a2[2 * k][i][r] * (float)hh[k][r] + a2[2 * k + 1][i][r] * (float)hh[k][4 + r];
        }
      }
      const float c1 = group4_sum(s1) * (1.f / 192.f) - bm1[i], c2 = group4_sum(s2) * (1.f / 192.f) - bm2[i];
      const float rr = 1.f / binv[i];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        f32x4 xo[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) xo[t][r] = rr * (a2[2 * k + t][i][r] - c1 - (float)hh[k][4 * t + r] * c2);
        db[i][k] = pack8(xo[0], xo[1]);
        if (mrow[i] < g.M) {
          *(float4*)(g.X + mo + tail_col(2 * k, 4 * lg)) = make_float4(xo[0][0], xo[0][1], xo[0][2], xo[0][3]);
          *(float4*)(g.X + mo + tail_col(2 * k + 1, 4 * lg)) = make_float4(xo[1][0], xo[1][1], xo[1][2], xo[1][3]);
          *(bf16x8*)(g.xb + mo + 32 * k + 8 * lg) = db[i][k];
        }
      }
    }
    if (!g.dO) return;
    // ---- proj dgrad: dO^T[d][row] = WprojT[d][:] . dXb[row][:] ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int e = 0; e < TPROJ; ++e) {
      const bf16* sb = lds + e * SLOT + lane * 8;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
          const bf16x8 wp = *(const bf16x8*)(sb + (6 * u + ks) * PIECE);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = mfma16(wp, db[i][ks], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) a2[4 * e + u][i] = acc[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (mrow[i] < g.M)
#pragma unroll
        for (int k = 0; k < 6; ++k) *(bf16x8*)(g.dO + (size_t)mrow[i] * D + 32 * k + 8 * lg) = pack8(a2[2 * k][i], a2[2 * k + 1][i]);
    return;
  }
  if constexpr (TAIL && KIND == 0) {
    // ---- a2 IS the updated residual stream (fp32, nothing staged through bf16): next LayerNorm in registers, 32- / 16-byte stores ----
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      float mean = 0.f, rs = 0.f;
      if (g.xhat) {
        float sum = 0.f;
#pragma unroll
        for (int ot = 0; ot < 12; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) sum += a2[ot][i][r];
        mean = group4_sum(sum) * (1.f / 192.f);
        float qs = 0.f;
#pragma unroll
        for (int ot = 0; ot < 12; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = a2[ot][i][r] - mean; qs += v * v; }
        rs = rsqrtf(group4_sum(qs) * (1.f / 192.f) + g.eps);
      }
      if (mrow[i] < g.M) {
        float* xp = g.X + (size_t)mrow[i] * D;
#pragma unroll
        for (int ot = 0; ot < 12; ++ot) {
          const f32x4 v = a2[ot][i];
          *(float4*)(xp + tail_col(ot, 4 * lg)) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (g.xhat) {
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            f32x4 lo, hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo[r] = (a2[2 * k][i][r] - mean) * rs; hi[r] = (a2[2 * k + 1][i][r] - mean) * rs; }
            *(bf16x8*)(g.xhat + (size_t)mrow[i] * D + 32 * k + 8 * lg) = pack8(lo, hi);
          }
          if (lg == 0) g.rstd[mrow[i]] = rs;
        }
      }
      if (g.qkv) {                  // the normalised rows as the B fragments of the next block's qkv projection
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          f32x4 lo, hi;
#pragma unroll
          for (int r = 0; r < 4; ++r) { lo[r] = (a2[2 * k][i][r] - mean) * rs; hi[r] = (a2[2 * k + 1][i][r] - mean) * rs; }
          xf[i][k] = pack8(lo, hi);
        }
      }
    }
    if (!g.qkv) return;
    // ---- the NEXT block's qkv projection (its norm1 affine is folded into the weight): 9 more ring entries of 4 output tiles; the
    // weight rows are permuted like proj's, so a lane's tile pair is 16 bytes of a qkv row ----
    if constexpr (PIPE) {
      int slot = (J0 + PENTRIES) % 3;
#pragma unroll 1
      for (int e = 0; e < TQKV; ++e) {
        const int J = J0 + PENTRIES + e;
        if (J < NTOT - 1) WAIT_DMA(2 * S); else wait_vm<2 * S>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (J + 2 < NTOT) dma(J + 2, slot == 0 ? 2 : slot - 1);
        const bf16* sb = lds + slot * SLOT + lane * 8;
        f32x4 acc[4][TPW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 b = *(const f32x4*)(s_bias + HID + 2 * D + 64 * e + tail_col(u, 4 * lg));
#pragma unroll
          for (int i = 0; i < TPW; ++i) acc[u][i] = b;
#pragma unroll
          for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 wq = *(const bf16x8*)(sb + (6 * u + ks) * PIECE);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[u][i] = mfma16(wq, xf[i][ks], acc[u][i]);
          }
        }
        // stores: at least S per iteration (the counted waits assume them); rows beyond M are sent out of range
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const unsigned off = mrow[i] < g.M ? (unsigned)mrow[i] * (3 * D * 2) + (64 * e + 32 * k + 8 * lg) * 2 : 0xFFFFFF00u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, pack8(acc[2 * k][i], acc[2 * k + 1][i])), r_qkv, off, 0, 0);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
    return;
  }
  if constexpr (TPW == 2) {
  // ---- epilogue: bf16(out [+ b2]) staged in LDS (aliases the ring: every wave must have left the loop) ----
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if MLP_DBG(g, 1) {                                          // timing ablation: no epilogue (keep the accumulators alive)
    if (a2[0][0][0] == 12345.678f && a2[11][1][3] == 1.f) g.X[0] = a2[5][0][1];
    return;
  }
  // Row-wise pass, 16 lanes per row (lane c16 holds elements {64 i + 4 c16 .. +3}), 8 passes of RP = 4 NW rows.  The residual rows of
  // passes 0-3 are requested BEFORE the staging writes and those of passes 4-7 before passes 0-3 are processed, so that the
  // pass is not a chain of eight dependent HBM round trips (the accumulators' 96 registers are free by then).
  constexpr int RP = 4 * NW;
  const int c16 = tid & 15, prow = tid >> 4;
  auto crow = [&](int pass) -> int { const int m = r0 + pass * RP + prow; return m < g.M ? m : g.M - 1; };
  auto owned = [&](int row, int m) -> bool { return row < g.rpw && m < g.M; };
  float4 xa[4][3], xb[4][3];
  bf16x4 ha[4][3], hb[4][3];       // backward: xhat2 rows
  float ra[4], rb[4];              // backward: rstd2
  auto fetch = [&](int pass, float4 (&xs)[3], bf16x4 (&hs)[3], float& rr) {
    const int mc = crow(pass);
    if (KIND && !g.X) {                  // round 4: bf16 residual gradient -- the incoming gradient IS the launch's dY operand (row-major (M,192))
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bf16x4 t = ((const bf16x4*)(g.xin + (size_t)mc * D))[16 * i + c16];
        xs[i] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) xs[i] = ((const float4*)(g.X + (size_t)mc * D))[16 * i + c16];
    }
    if (KIND) {
#pragma unroll
      for (int i = 0; i < 3; ++i) hs[i] = ((const bf16x4*)(g.xhat + (size_t)mc * D))[16 * i + c16];
      rr = g.rstd[mc];
    }
  };
#pragma unroll
  for (int p = 0; p < 4; ++p) fetch(p, xa[p], ha[p], ra[p]);
  bf16* Cs = lds;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ot = 0; ot < 12; ++ot) {
      f32x4 v = a2[ot][i];
      if (!KIND) {
        const f32x4 bb = *(const f32x4*)(s_bias + HID + 16 * ot + 4 * lg);
        v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3];
      }
      *(bf16x4*)(Cs + (16 * NW * i + 16 * w + l15) * CSTR + 16 * ot + 4 * lg) = pack4(v);
    }
  barrier_lds();
#pragma unroll
  for (int p = 0; p < 4; ++p) fetch(4 + p, xb[p], hb[p], rb[p]);
  // forward: X += branch; next LayerNorm.  backward: dX += LayerNorm-backward(dxhat); dXb = bf16(dX).  All arithmetic first, all
  // stores last (DESIGN.md "packed-fp32 hazard": the library also carries no packed-fp32 VALU instruction).
  auto finish = [&](int pass, float4 (&xs)[3], const bf16x4 (&hs)[3], float rr) {
    const int row = pass * RP + prow;
    const int m = r0 + row;
    float4* xp = (float4*)(g.X + (size_t)crow(pass) * D);
    float v[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bf16x4 t = *(const bf16x4*)(Cs + row * CSTR + 64 * i + 4 * c16);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * i + e] = (float)t[e];
    }
    if (KIND) {
      // dX += rstd (v - mean(v) - xhat mean(v xhat)); the affine is already folded into W1T (same arithmetic and summation
      // order as gemm_ws_kernel's EPI_LNBWD epilogue)
      float h[12];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[4 * i + e] = (float)hs[i][e]; s1 += v[4 * i + e]; s2 += v[4 * i + e] * h[4 * i + e]; }
      const float c1 = wave_sum16(s1) * (1.f / 192.f), c2 = wave_sum16(s2) * (1.f / 192.f);
      bf16x4 bq[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float4 x = xs[i];
        x.x += rr * (v[4 * i] - c1 - h[4 * i] * c2);
        x.y += rr * (v[4 * i + 1] - c1 - h[4 * i + 1] * c2);
        x.z += rr * (v[4 * i + 2] - c1 - h[4 * i + 2] * c2);
        x.w += rr * (v[4 * i + 3] - c1 - h[4 * i + 3] * c2);
        xs[i] = x;
        f32x4 t = {x.x, x.y, x.z, x.w};
        bq[i] = pack4(t);
      }
      if (owned(row, m)) {
        bf16x4* bp = (bf16x4*)(g.xb + (size_t)m * D);
        if (g.X && !MLP_DBG(g, 64)) {
#pragma unroll
          for (int i = 0; i < 3; ++i) xp[16 * i + c16] = xs[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) bp[16 * i + c16] = bq[i];
      }
      return;
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float4 x = xs[i];
      x.x += v[4 * i]; x.y += v[4 * i + 1]; x.z += v[4 * i + 2]; x.w += v[4 * i + 3];
      xs[i] = x;
      v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
      sum += (x.x + x.y) + (x.z + x.w);
    }
    if (g.xhat) {
      const float mean = wave_sum16(sum) * (1.f / 192.f);
      float qs = 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) { v[e] -= mean; qs += v[e] * v[e]; }
      const float r = rsqrtf(wave_sum16(qs) * (1.f / 192.f) + g.eps);
      bf16x4 hq[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        f32x4 t = {v[4 * i] * r, v[4 * i + 1] * r, v[4 * i + 2] * r, v[4 * i + 3] * r};
        hq[i] = pack4(t);
      }
      if (owned(row, m)) {
        bf16x4* hp = (bf16x4*)(g.xhat + (size_t)m * D);
#pragma unroll
        for (int i = 0; i < 3; ++i) xp[16 * i + c16] = xs[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) hp[16 * i + c16] = hq[i];
        if (c16 == 0) g.rstd[m] = r;
      }
    } else if (owned(row, m)) {
#pragma unroll
      for (int i = 0; i < 3; ++i) xp[16 * i + c16] = xs[i];
    }
  };
#pragma unroll
  for (int p = 0; p < 4; ++p) finish(p, xa[p], ha[p], ra[p]);
#pragma unroll
  for (int p = 0; p < 4; ++p) finish(4 + p, xb[p], hb[p], rb[p]);
  }
#undef WAIT_DMA
}

// Weight stream of one block: chunk c (hidden units 32 c .. 32 c + 31) = 24 pieces of 64 x 16 bytes;
//   piece 6 t + ks (t = 0, 1; ks = 0..5): lane (l15, lg) = W1f[32 c + 8 (l15 >> 2) + 4 t + (l15 & 3)][32 ks + 8 lg .. +7]
//   piece 12 + ot  (ot = 0..11):         lane (l15, lg) = W2[16 ot + l15][32 c + 8 lg .. +7]
struct MlpPrepArgs { const char* base; size_t blk0, stride, off_w1, off_w2, off_out, off_wp; int bwd; size_t off_wq; int n_next; };   // off_wq: the NEXT block's folded qkv weight (forward tail), or NO_WP     // off_wp: proj weight (bwd: its transpose), or NO_WP
constexpr size_t NO_WP = ~(size_t)0;
// The stream buffer holds THREE images and the GELU table: the plain one (NCHUNK entries), the SKEWED one of the pipelined forward
// (PENTRIES entries: entry j = fc1 fragments of chunk j | fc2 fragments of chunk j - PSKEW; the missing halves are zeros) and the
// BLOCK-TAIL one (TENTRIES entries: TPROJ entries of proj-weight fragments, output tile ot = rows tail_col(ot, .) of the weight, then the
// skewed image with the rows of the fc2 fragments permuted the same way; zeros when no proj weight is given).
constexpr int STREAM_ENTRIES = NCHUNK + PENTRIES + TENTRIES;
__global__ __launch_bounds__(256) void mlp_stream_prep_kernel(const MlpPrepArgs a) {
  const int blk = blockIdx.y;
  const char* q = a.base + a.blk0 + (size_t)blk * a.stride;
  const bf16* w1 = (const bf16*)(q + a.off_w1);
  const bf16* w2 = (const bf16*)(q + a.off_w2);
  bf16* out = (bf16*)(const_cast<char*>(q) + a.off_out);
  const int e = blockIdx.x * 256 + threadIdx.x;          // 16-byte element of the stream: STREAM_ENTRIES * 24 * 64
  if (e >= STREAM_ENTRIES * CH_PIECES * 64) return;
  const int lane = e & 63, piece = (e >> 6) % CH_PIECES, entry = e / (64 * CH_PIECES);
  const int l15 = lane & 15, lg = lane >> 4;
  bf16x8 v = {};
  const bf16* src = nullptr;
  if (entry >= NCHUNK + PENTRIES) {                       // block-tail image
    const int t = entry - (NCHUNK + PENTRIES);
    if (a.off_wp != NO_WP && a.bwd) {
      // backward block tail: NCHUNK plain entries whose second-GEMM (fc1 dgrad) rows are permuted by tail_col, then TPROJ entries of
      // WprojT fragments (the proj dgrad behind the norm2 backward)
      const bf16* wp = (const bf16*)(q + a.off_wp);
      if (t < NCHUNK) {
        if (piece < 12) {
          const int tt = piece / 6, ks = piece - 6 * tt;
          src = w1 + (size_t)(HC * t + 8 * (l15 >> 2) + 4 * tt + (l15 & 3)) * D + 32 * ks + 8 * lg;
        } else {
          src = w2 + (size_t)tail_col(piece - 12, l15) * HID + HC * t + 8 * lg;
        }
      } else if (t < NCHUNK + TPROJ) {
        const int ot = 4 * (t - NCHUNK) + piece / 6, ks = piece % 6;
        src = wp + (size_t)tail_col(ot, l15) * D + 32 * ks + 8 * lg;
      }
    } else if (a.off_wp != NO_WP) {
      if (t < TPROJ) {
        const bf16* wp = (const bf16*)(q + a.off_wp);
        const int ot = 4 * t + piece / 6, ks = piece % 6;
        src = wp + (size_t)tail_col(ot, l15) * D + 32 * ks + 8 * lg;
      } else if (t >= TPROJ + PENTRIES) {                  // the next block's qkv weight (none behind the last block)
        if (a.off_wq != NO_WP && (int)blockIdx.y < a.n_next) {             // n_next: blocks (from 0) that have a next block
          const bf16* wq = (const bf16*)(q + a.off_wq);
          const int ot = 4 * (t - TPROJ - PENTRIES) + piece / 6, ks = piece % 6;
          src = wq + (size_t)tail_col(ot, l15) * D + 32 * ks + 8 * lg;
        }
      } else {
        const int c = t - TPROJ - (piece < 12 ? 0 : PSKEW);
        if (c >= 0 && c < NCHUNK) {
          if (piece < 12) {
            const int tt = piece / 6, ks = piece - 6 * tt;
            src = w1 + (size_t)(HC * c + 8 * (l15 >> 2) + 4 * tt + (l15 & 3)) * D + 32 * ks + 8 * lg;
          } else {
            src = w2 + (size_t)tail_col(piece - 12, l15) * HID + HC * c + 8 * lg;
          }
        }
      }
    }
  } else {
    int c = entry;                                        // hidden chunk this piece belongs to
    if (entry >= NCHUNK) c = entry - NCHUNK - (piece < 12 ? 0 : PSKEW);
    if (c >= 0 && c < NCHUNK) {
      if (piece < 12) {
        const int t = piece / 6, ks = piece - 6 * t;
        src = w1 + (size_t)(HC * c + 8 * (l15 >> 2) + 4 * t + (l15 & 3)) * D + 32 * ks + 8 * lg;
      } else {
        src = w2 + (size_t)(16 * (piece - 12) + l15) * HID + HC * c + 8 * lg;
      }
    }
  }
  if (src) v = *(const bf16x8*)src;
  *(bf16x8*)(out + (size_t)e * 8) = v;
}

// the GELU table (see GT_*): entry = bf16 gelu | bf16 gelu' << 16 of the bf16 input pattern, computed by gelu_and_grad itself
__global__ __launch_bounds__(256) void mlp_gelu_table_kernel(const MlpPrepArgs a) {
  const char* q = a.base + a.blk0 + (size_t)blockIdx.y * a.stride;
  unsigned* out = (unsigned*)(const_cast<char*>(q) + a.off_out + (size_t)STREAM_ENTRIES * CH_ELEMS * sizeof(bf16));
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= GT_ENTRIES) return;
  const int k = e & (GT_NEG - 1);
  unsigned v = 0;
  if (k < GT_N) {
    const unsigned short u = (unsigned short)((e >= GT_NEG ? 0x8000 : 0) | (GT_EM_LO + k));
    const bf16 x = __builtin_bit_cast(bf16, u);
    float ga, gd;
    gelu_and_grad((float)x, ga, gd);
    const bf16x2 pr = {(bf16)ga, (bf16)gd};
    v = __builtin_bit_cast(unsigned, pr);
  }
  out[e] = v;
}

}  // namespace

// Schedule of the fused MLP forward.  The product runs ONE: 8 waves, in-wave software pipeline (fc1 of chunk j under the GELU of chunk
// j - 1) with the GELU looked up in an LDS table of the bf16 input patterns.  Measured against it on MI355X, M = 50 432 (developer
// library, ROVIT_KNOB_MLP_SCHEDULE; tools/mlp_ablate.py, profiles/r03_mlp_ablation.json): 8 = 8 waves in lockstep, exact-erf GELU in the
// loop (75.0 against 64.2 us per training launch), 9 = waves 4-7 staggered half a chunk behind waves 0-3 (68 against 71 us lockstep
// on one box, 76 against 75 on another: the two extra barriers per chunk give back what the hidden GELU gains), 4 = two 128-row
// workgroups of four waves per CU (each streams the whole 590 KB: step 5.97 against 5.86 ms).
static int mlp_schedule() { return ROVIT_KNOB(ROVIT_KNOB_MLP_SCHEDULE, 10); }
// token rows per workgroup of the 8-wave kernels: a launch lasts as long as ONE workgroup (197 workgroups of 256 rows at batch 256 run
// side by side on 256 CUs), so while everything fits one round, 240 rows per workgroup (211 workgroups, one idle row tile each) shorten
// it: step 4.76 -> 4.69 ms, batch-256 inference 1.30 -> 1.28 ms; from two rounds on (batch 512: 2.26 -> 2.33 ms) full workgroups win.
static int mlp_rpw(long total_rows) {
  const int forced = ROVIT_KNOB(ROVIT_KNOB_MLP_RPW, 0);
  if (forced >= 16 && forced <= 256) return forced;
  return (total_rows + 239) / 240 <= 256 ? 240 : 256;
}

extern "C" size_t rovit_mlp_stream_bytes(void) { return (size_t)STREAM_ENTRIES * CH_ELEMS * sizeof(bf16) + GT_ENTRIES * sizeof(unsigned); }

// (internal) streams of `depth` blocks laid out inside the prepared-weight buffer of rovit_vit_prepare
int rovit_mlp_stream_prep_blocks(const void* prep_base, size_t blk0, size_t stride, size_t off_w1, size_t off_w2, size_t off_out,
                                 size_t off_wp, int bwd, size_t off_wq_next, int depth, rovit_stream_t stream, bool gelu_tables) {
  const MlpPrepArgs a{(const char*)prep_base, blk0, stride, off_w1, off_w2, off_out, off_wp, bwd, off_wq_next, depth - 1};
  hipLaunchKernelGGL(mlp_stream_prep_kernel, dim3((STREAM_ENTRIES * CH_PIECES * 64 + 255) / 256, depth), dim3(256), 0, (hipStream_t)stream, a);
  // only the forward looks GELU up; the table does not depend on the weights: a caller that prepares the same buffer again may skip it
  if (!bwd && gelu_tables) hipLaunchKernelGGL(mlp_gelu_table_kernel, dim3(GT_ENTRIES / 256, depth), dim3(256), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("mlp_stream_prep_kernel");
  return ROVIT_OK;
}

// w1f: bf16 (768,192) fc1 weight with the LayerNorm affine folded in (rovit_prep_weight's Wf); w2: bf16 (192,768) fc2 weight;
// wstream: rovit_mlp_stream_bytes() bytes, 16-byte aligned.
static int mlp_prepare_stream_impl(const void* w1f, const void* w2, const void* wproj, void* wstream, rovit_stream_t stream, int bwd = 0,
                                   const void* wqkv_next = nullptr);
extern "C" int rovit_mlp_prepare_stream(const void* w1f, const void* w2, void* wstream, rovit_stream_t stream) {
  return mlp_prepare_stream_impl(w1f, w2, nullptr, wstream, stream);
}
#ifdef ROVIT_DEV
// the dgrad image with the backward block tail (rovit_block_tail_bwd): w1f := W2T (768,192), w2 := W1T folded (192,768), wprojT = the
// TRANSPOSED bf16 proj weight (192,192), row d = the weights of attention-output column d
extern "C" int rovit_mlp_prepare_stream_tail_bwd(const void* w2T, const void* w1T, const void* wprojT, void* wstream, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(wprojT && rovit_aligned16(wprojT), ROVIT_ERR_NULL, "mlp_prepare_stream_tail_bwd: wprojT missing or misaligned");
  return mlp_prepare_stream_impl(w2T, w1T, wprojT, wstream, stream, 1);
}
#endif
// ... with the block-tail image too (rovit_block_tail_fwd): wproj = the bf16 attention-output projection weight (192,192)
// wqkv_next (may be NULL): the bf16 qkv weight (576,192) of the NEXT block with its norm1 affine folded in -- rovit_block_tail_fwd then
// also writes that block's qkv projection
extern "C" int rovit_mlp_prepare_stream_tail(const void* w1f, const void* w2, const void* wproj, const void* wqkv_next, void* wstream,
                                             rovit_stream_t stream) {
  ROVIT_CHECK_ARG(wproj && rovit_aligned16(wproj) && rovit_aligned16(wqkv_next), ROVIT_ERR_NULL, "mlp_prepare_stream_tail: wproj missing or misaligned");
  return mlp_prepare_stream_impl(w1f, w2, wproj, wstream, stream, 0, wqkv_next);
}
static int mlp_prepare_stream_impl(const void* w1f, const void* w2, const void* wproj, void* wstream, rovit_stream_t stream, int bwd,
                                   const void* wqkv_next) {
  ROVIT_CHECK_ARG(w1f && w2 && wstream, ROVIT_ERR_NULL, "mlp_prepare_stream: null pointer");
  ROVIT_CHECK_ARG(rovit_aligned16(w1f) && rovit_aligned16(w2) && rovit_aligned16(wstream), ROVIT_ERR_ALIGN, "mlp_prepare_stream: alignment");
  // one "block" whose fields are addressed relative to w1f
  const MlpPrepArgs a{(const char*)w1f, 0, 0, 0, (size_t)((const char*)w2 - (const char*)w1f), (size_t)((char*)wstream - (const char*)w1f),
                      wproj ? (size_t)((const char*)wproj - (const char*)w1f) : NO_WP, bwd,
                      wqkv_next ? (size_t)((const char*)wqkv_next - (const char*)w1f) : NO_WP, wqkv_next ? 1 : 0};
  hipLaunchKernelGGL(mlp_stream_prep_kernel, dim3((STREAM_ENTRIES * CH_PIECES * 64 + 255) / 256, 1), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(mlp_gelu_table_kernel, dim3(GT_ENTRIES / 256, 1), dim3(256), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("mlp_stream_prep_kernel");
  return ROVIT_OK;
}

// X(M,192) += fc2(GELU(fc1(xhat2))) and the LayerNorm behind it, one launch.  act / dact: NULL for inference (nothing kept);
// dact alone may be NULL (gelu' recomputed by the backward).  xhat_out NULL: no LayerNorm.
// act / dact are CHUNK-MAJOR: element (row m, hidden unit h) of a tensor of act_rows rows lives at ((h / 32) * act_rows + m) * 32 + h % 32.
// A launch may cover a row range of such a tensor: pass the pointers advanced by first_row * 32 elements and act_rows of the whole.
extern "C" int rovit_mlp_fused_fwd(const void* xhat2, const void* wstream, const float* b1, const float* b2, void* act, void* dact,
                                   float* X, void* xhat_out, float* rstd_out, float eps, int M, int act_rows, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(xhat2 && wstream && b1 && b2 && X, ROVIT_ERR_NULL, "mlp_fused_fwd: null pointer");
  ROVIT_CHECK_ARG(M > 0 && act_rows >= M && (size_t)act_rows * HID * 2 < ((size_t)1 << 31), ROVIT_ERR_SHAPE,
                  "mlp_fused_fwd: M = %d, act_rows = %d out of range", M, act_rows);
  ROVIT_CHECK_ARG(act || !dact, ROVIT_ERR_NULL, "mlp_fused_fwd: dact without act");
  ROVIT_CHECK_ARG(!xhat_out || rstd_out, ROVIT_ERR_NULL, "mlp_fused_fwd: rstd_out missing");
  ROVIT_CHECK_ARG(rovit_aligned16(xhat2) && rovit_aligned16(wstream) && rovit_aligned16(b1) && rovit_aligned16(b2) && rovit_aligned16(X) &&
                      rovit_aligned16(act) && rovit_aligned16(dact) && rovit_aligned16(xhat_out),
                  ROVIT_ERR_ALIGN, "mlp_fused_fwd: buffers must be 16-byte aligned");
  MlpArgs g{};
  g.xin = (const bf16*)xhat2; g.wstream = (const bf16*)wstream; g.b1 = b1; g.b2 = b2; g.act = (bf16*)act; g.dact = (bf16*)dact;
  g.X = X; g.xhat = (bf16*)xhat_out; g.rstd = rstd_out; g.eps = eps; g.M = M; g.act_rows = act_rows;
#ifdef ROVIT_DEV
  g.dbg = ROVIT_KNOB(ROVIT_KNOB_MLP_DBG, 0);
#endif
  hipStream_t st = (hipStream_t)stream;
  const int sched = mlp_schedule();
  const int nw = sched == 4 ? 4 : 8;
  g.rpw = nw == 8 ? mlp_rpw(act_rows) : 32 * nw;
  const dim3 grid((M + g.rpw - 1) / g.rpw), block(64 * nw);
  if (sched == 10) {
    g.gelu_table = g.wstream + (size_t)STREAM_ENTRIES * CH_ELEMS;
    g.wstream += (size_t)NCHUNK * CH_ELEMS;                // the skewed image
#define LAUNCH_PIPE(MD)                                                                                                          \
  do {                                                                                                                           \
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<0, MD, 8, false, true>, lds_bytes(0, 8) + GT_ENTRIES * 4),   \
                    ROVIT_ERR_LAUNCH, "mlp_fused_fwd: cannot raise the LDS limit");                                              \
    hipLaunchKernelGGL((mlp_fused_kernel<0, MD, 8, false, true>), grid, block, lds_bytes(0, 8) + GT_ENTRIES * 4, st, g);         \
  } while (0)
    if (!act) LAUNCH_PIPE(0); else if (!dact) LAUNCH_PIPE(1); else LAUNCH_PIPE(2);
#undef LAUNCH_PIPE
  }
#ifdef ROVIT_DEV          // the schedules that lost (see mlp_schedule): A/B in the developer library only
#define LAUNCH_MODE(MD, NWV)                                                                                                     \
  do {                                                                                                                           \
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<0, MD, NWV>, lds_bytes(0, NWV)), ROVIT_ERR_LAUNCH,           \
                    "mlp_fused_fwd: cannot raise the LDS limit");                                                                \
    hipLaunchKernelGGL((mlp_fused_kernel<0, MD, NWV>), grid, block, lds_bytes(0, NWV), st, g);                                   \
  } while (0)
  else if (sched == 9) {
#define LAUNCH_STAG(MD)                                                                                                          \
  do {                                                                                                                           \
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<0, MD, 8, true>, lds_bytes(0, 8) + 8 * 6 * 1024), ROVIT_ERR_LAUNCH, \
                    "mlp_fused_fwd: cannot raise the LDS limit");                                                                \
    hipLaunchKernelGGL((mlp_fused_kernel<0, MD, 8, true>), grid, block, lds_bytes(0, 8) + 8 * 6 * 1024, st, g);                  \
  } while (0)
    if (!act) LAUNCH_STAG(0); else if (!dact) LAUNCH_STAG(1); else LAUNCH_STAG(2);
#undef LAUNCH_STAG
  } else if (sched == 8) {
    if (!act) LAUNCH_MODE(0, 8); else if (!dact) LAUNCH_MODE(1, 8); else LAUNCH_MODE(2, 8);
  } else if (sched == 4) {
    if (!act) LAUNCH_MODE(0, 4); else if (!dact) LAUNCH_MODE(1, 4); else LAUNCH_MODE(2, 4);
  }
#undef LAUNCH_MODE
#endif
  else {
    rovit_set_error("mlp_fused_fwd: unknown schedule %d", sched);
    return ROVIT_ERR_SHAPE;
  }
  ROVIT_CHECK_LAUNCH("mlp_fused_kernel (forward)");
  return ROVIT_OK;
}

// Everything of a block behind the attention in ONE launch ("block tail"; timm Block: x = x + proj(attn); x = x + mlp(norm2(x)); then
// the next block's norm1 -- models/backbone.py:23-25):
//   X (M,192) += o Wp^T + bp;  xhat2 / rstd2 = LayerNorm(X) (kept for the backward; NULL: inference);
//   X += fc2(GELU(fc1(xhat2)));  xhat_out / rstd_out = LayerNorm(X) (NULL: none).
// o: bf16 (M,192) attention output; wstream: rovit_mlp_prepare_stream_tail's buffer; act / dact as rovit_mlp_fused_fwd (chunk-major).
// The residual stream stays in fp32 registers between the two halves: nothing is staged through bf16, X is read and written once.
extern "C" int rovit_block_tail_fwd(const void* o, const void* wstream, const float* bp, const float* b1, const float* b2, float* X,
                                    void* xhat2, float* rstd2, void* act, void* dact, void* xhat_out, float* rstd_out, const float* bq_next,
                                    void* qkv_next, float eps, int M, int act_rows, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(!qkv_next || (bq_next && xhat_out && rovit_aligned16(bq_next) && rovit_aligned16(qkv_next)), ROVIT_ERR_NULL,
                  "block_tail_fwd: the qkv phase needs bq_next, xhat_out and aligned buffers");
  ROVIT_CHECK_ARG((size_t)M * 3 * D * 2 < ((size_t)1 << 31), ROVIT_ERR_SHAPE, "block_tail_fwd: M = %d out of range", M);
  ROVIT_CHECK_ARG(o && wstream && bp && b1 && b2 && X, ROVIT_ERR_NULL, "block_tail_fwd: null pointer");
  ROVIT_CHECK_ARG(M > 0 && act_rows >= M && (size_t)act_rows * HID * 2 < ((size_t)1 << 31), ROVIT_ERR_SHAPE,
                  "block_tail_fwd: M = %d, act_rows = %d out of range", M, act_rows);
  ROVIT_CHECK_ARG(act || !dact, ROVIT_ERR_NULL, "block_tail_fwd: dact without act");
  ROVIT_CHECK_ARG((!xhat_out || rstd_out) && (!xhat2 || rstd2), ROVIT_ERR_NULL, "block_tail_fwd: rstd output missing");
  ROVIT_CHECK_ARG(rovit_aligned16(o) && rovit_aligned16(wstream) && rovit_aligned16(bp) && rovit_aligned16(b1) && rovit_aligned16(b2) &&
                      rovit_aligned16(X) && rovit_aligned16(xhat2) && rovit_aligned16(act) && rovit_aligned16(dact) && rovit_aligned16(xhat_out),
                  ROVIT_ERR_ALIGN, "block_tail_fwd: buffers must be 16-byte aligned");
  MlpArgs g{};
  g.xin = (const bf16*)o; g.b1 = b1; g.b2 = b2; g.bp = bp; g.act = (bf16*)act; g.dact = (bf16*)dact; g.xhat2 = (bf16*)xhat2; g.rstd2 = rstd2;
  g.X = X; g.xhat = (bf16*)xhat_out; g.rstd = rstd_out; g.eps = eps; g.M = M; g.act_rows = act_rows; g.bq = bq_next; g.qkv = (bf16*)qkv_next;
  g.gelu_table = (const bf16*)wstream + (size_t)STREAM_ENTRIES * CH_ELEMS;
  g.wstream = (const bf16*)wstream + (size_t)(NCHUNK + PENTRIES) * CH_ELEMS;          // the block-tail image
  g.rpw = mlp_rpw(act_rows);
  const dim3 grid((M + g.rpw - 1) / g.rpw);
#ifdef ROVIT_DEV        // measured slower (same box: 97.5 against 92.6 us training, 87.9 against 84.0 inference): developer library only
  if (ROVIT_KNOB(ROVIT_KNOB_TAIL_WAVES, TAIL_WAVES_DEFAULT) == 16) {
    // sixteen waves of one 16-row tile (four waves per SIMD): see the kernel's TPW note
    const size_t lds = (size_t)NSLOT * slot_elems(0, 16) * sizeof(bf16) + (HID + D + 4 * D) * sizeof(float) + GT_ENTRIES * 4;
#define LAUNCH_TAIL16(MD)                                                                                                        \
  do {                                                                                                                           \
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<0, MD, 16, false, true, true, 1>, lds), ROVIT_ERR_LAUNCH,    \
                    "block_tail_fwd: cannot raise the LDS limit");                                                               \
    hipLaunchKernelGGL((mlp_fused_kernel<0, MD, 16, false, true, true, 1>), grid, dim3(1024), lds, (hipStream_t)stream, g);      \
  } while (0)
    if (!act) LAUNCH_TAIL16(0); else if (!dact) LAUNCH_TAIL16(1); else LAUNCH_TAIL16(2);
#undef LAUNCH_TAIL16
    ROVIT_CHECK_LAUNCH("mlp_fused_kernel (block tail, 16 waves)");
    return ROVIT_OK;
  }
#endif
  const size_t lds = lds_bytes(0, 8) + 4 * D * sizeof(float) + GT_ENTRIES * 4;
  const dim3 block(512);
#define LAUNCH_TAIL(MD)                                                                                                          \
  do {                                                                                                                           \
    ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<0, MD, 8, false, true, true>, lds), ROVIT_ERR_LAUNCH,        \
                    "block_tail_fwd: cannot raise the LDS limit");                                                               \
    hipLaunchKernelGGL((mlp_fused_kernel<0, MD, 8, false, true, true>), grid, block, lds, (hipStream_t)stream, g);               \
  } while (0)
  if (!act) LAUNCH_TAIL(0); else if (!dact) LAUNCH_TAIL(1); else LAUNCH_TAIL(2);
#undef LAUNCH_TAIL
  ROVIT_CHECK_LAUNCH("mlp_fused_kernel (block tail)");
  return ROVIT_OK;
}

// (dact and dpre are chunk-major [24][M][32], like the forward's act / dact with act_rows = M.)
// The dgrad chain of the MLP half in one launch: dpre (M,768) = (dY W2T^T) * dact, kept for the fc1 weight gradient;
// dX (M,192) += LayerNorm-2-backward(dpre W1T^T) with xhat2 / rstd2 of the forward; dXb = bf16(dX).  wstream_bwd: the image
// rovit_mlp_prepare_stream builds from (w1f := W2T (768,192), w2 := W1T folded (192,768)).
extern "C" int rovit_mlp_fused_bwd(const void* dY, const void* wstream_bwd, const void* dact, void* dpre, const void* xhat2,
                                   const float* rstd2, float* dX, void* dXb, int M, rovit_stream_t stream) {
  // dX == NULL (round 4): the residual gradient travels in bf16 -- dXb = bf16(float(dY) + LayerNorm-2-backward(...)), no fp32 dX read or written
  ROVIT_CHECK_ARG(dY && wstream_bwd && dact && dpre && xhat2 && rstd2 && dXb, ROVIT_ERR_NULL, "mlp_fused_bwd: null pointer");
  ROVIT_CHECK_ARG(M > 0 && (size_t)M * HID * 2 < ((size_t)1 << 31), ROVIT_ERR_SHAPE, "mlp_fused_bwd: M = %d out of range", M);
  ROVIT_CHECK_ARG(rovit_aligned16(dY) && rovit_aligned16(wstream_bwd) && rovit_aligned16(dact) && rovit_aligned16(dpre) &&
                      rovit_aligned16(xhat2) && rovit_aligned16(dX) && rovit_aligned16(dXb),
                  ROVIT_ERR_ALIGN, "mlp_fused_bwd: buffers must be 16-byte aligned");
  MlpArgs g{};
  g.xin = (const bf16*)dY; g.wstream = (const bf16*)wstream_bwd; g.act = (bf16*)dpre; g.mul = (const bf16*)dact; g.X = dX;
  g.xhat = (bf16*)const_cast<void*>(xhat2); g.rstd = const_cast<float*>(rstd2); g.xb = (bf16*)dXb; g.M = M;
#ifdef ROVIT_DEV
  g.dbg = ROVIT_KNOB(ROVIT_KNOB_MLP_DBG, 0) | (ROVIT_KNOB(ROVIT_KNOB_SKIP_DX_FP32_STORE, 0) ? 64 : 0);
#endif
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<1, 1, 8>, lds_bytes(1, 8)), ROVIT_ERR_LAUNCH,
                  "mlp_fused_bwd: cannot raise the LDS limit");
  g.rpw = mlp_rpw(M);
  hipLaunchKernelGGL((mlp_fused_kernel<1, 1, 8>), dim3((M + g.rpw - 1) / g.rpw), dim3(512), lds_bytes(1, 8), (hipStream_t)stream, g);
  ROVIT_CHECK_LAUNCH("mlp_fused_kernel (backward)");
  return ROVIT_OK;
}

#ifdef ROVIT_DEV      // round 3's backward block tail: measured slower in the step (4.87 against 4.78 ms), developer library only
// rovit_mlp_fused_bwd with the norm2 backward in registers (fp32 dxhat2, nothing staged through bf16) and, behind it, the proj dgrad
// dO (M,192) = dXb Wproj in the same launch (dO NULL: none).  wstream_bwd from rovit_mlp_prepare_stream_tail_bwd.
extern "C" int rovit_block_tail_bwd(const void* dY, const void* wstream_bwd, const void* dact, void* dpre, const void* xhat2,
                                    const float* rstd2, float* dX, void* dXb, void* dO, int M, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(dY && wstream_bwd && dact && dpre && xhat2 && rstd2 && dX && dXb, ROVIT_ERR_NULL, "block_tail_bwd: null pointer");
  ROVIT_CHECK_ARG(M > 0 && (size_t)M * HID * 2 < ((size_t)1 << 31), ROVIT_ERR_SHAPE, "block_tail_bwd: M = %d out of range", M);
  ROVIT_CHECK_ARG(rovit_aligned16(dY) && rovit_aligned16(wstream_bwd) && rovit_aligned16(dact) && rovit_aligned16(dpre) &&
                      rovit_aligned16(xhat2) && rovit_aligned16(dX) && rovit_aligned16(dXb) && rovit_aligned16(dO),
                  ROVIT_ERR_ALIGN, "block_tail_bwd: buffers must be 16-byte aligned");
  MlpArgs g{};
  g.xin = (const bf16*)dY; g.act = (bf16*)dpre; g.mul = (const bf16*)dact; g.X = dX; g.dO = (bf16*)dO;
  g.wstream = (const bf16*)wstream_bwd + (size_t)(NCHUNK + PENTRIES) * CH_ELEMS;      // the backward block-tail image
  g.xhat = (bf16*)const_cast<void*>(xhat2); g.rstd = const_cast<float*>(rstd2); g.xb = (bf16*)dXb; g.M = M;
#ifdef ROVIT_DEV
  g.dbg = ROVIT_KNOB(ROVIT_KNOB_MLP_DBG, 0) | (ROVIT_KNOB(ROVIT_KNOB_SKIP_DX_FP32_STORE, 0) ? 64 : 0);
#endif
  ROVIT_CHECK_ARG(rovit_set_max_lds((const void*)mlp_fused_kernel<1, 1, 8, false, false, true>, lds_bytes(1, 8)), ROVIT_ERR_LAUNCH,
                  "block_tail_bwd: cannot raise the LDS limit");
  g.rpw = mlp_rpw(M);
  hipLaunchKernelGGL((mlp_fused_kernel<1, 1, 8, false, false, true>), dim3((M + g.rpw - 1) / g.rpw), dim3(512), lds_bytes(1, 8), (hipStream_t)stream, g);
  ROVIT_CHECK_LAUNCH("mlp_fused_kernel (backward block tail)");
  return ROVIT_OK;
}
#endif
