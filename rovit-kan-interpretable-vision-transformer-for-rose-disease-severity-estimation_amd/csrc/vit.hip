// DeiT-Tiny backbone forward / backward: the launch sequence over the kernels in gemm.hip, attention.hip and
// elementwise.hip.  Work is ordered on the caller's stream; ONE library-owned side stream is forked from and joined back into
// it inside each call (the second half-batch chain in the forward of batches >= 16, the weight gradients in the
// backward).  Nothing here allocates device memory or synchronises with the host.
//
// Reference being restated: DeiTTinyBackbone.forward (/root/reference/models/backbone.py:23-25) ->
// timm VisionTransformer.forward, and its autograd backward (training/trainer.py:119,136).
//
// Data layout in HBM (B images, T = 197 tokens, M = B*T rows):
//   X      fp32 (M,192)   residual stream, updated in place by the proj / fc2 epilogues
//   per block, kept for backward when training: xhat1/xhat2 bf16 (M,192) + rstd (M), qkv bf16 (M,576),
//   attention out bf16 (M,192) + lse2 (B,3,T), act = gelu(pre) and dact = gelu'(pre) bf16 (M,768)
//   LayerNorm affines are folded into the following Linear (W*gamma, b + W beta) by rovit_vit_prepare, so the
//   GEMM operand is the normalised xhat itself and the wgrad recovers dgamma/dbeta from G = dY^T xhat.
#include <algorithm>
#include <vector>

#include "common.h"

namespace {

constexpr int T = 197, D = 192, H = 3, MLP = 768, PD = 768;
enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_MUL = 3, EPI_PATCH = 4 };
enum { P_CLS = 0, P_POS, P_PATCH_W, P_PATCH_B, P_NORM_W, P_NORM_B, P_BLOCK0 };
enum { B_N1W = 0, B_N1B, B_QKVW, B_QKVB, B_PROJW, B_PROJB, B_N2W, B_N2B, B_FC1W, B_FC1B, B_FC2W, B_FC2B, B_COUNT };

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// Which MLP half a call runs -- an ARGUMENT of rovit_vit_forward / rovit_vit_backward (`mlp_path`), not library state: the forward, the
// dgrad chain and the weight gradients of a training step must agree on it, because the one-launch kernels (mlp_fused.hip) keep act,
// gelu' and dpre CHUNK-MAJOR ([24][M][32]) and the two-launch kernels row-major (RovitWgradDesc::a_blk / y_blk tells the weight-gradient
// launch which).  ROVIT_MLP_AUTO picks by size: a fused launch has one workgroup per 240 / 256 token rows, so below 34 000 rows (batch 173)
// the two-launch kernels, whose grids also split the output columns, fill the chip better.  Measured crossover on MI355X
// (tools/fwd_small_batch.py, bench.py --batch n): batch 176 for the inference forward (1.52 ms either way; batch 128: 1.33 fused against
// 1.23 ms, batch 1: 0.99 against 0.66 ms) and for the training step alike.
constexpr long MLP_FUSED_MIN_ROWS = 34000;
inline bool mlp_one_launch(int mlp_path, long rows) {
  return mlp_path == ROVIT_MLP_ONE_LAUNCH || (mlp_path == ROVIT_MLP_AUTO && rows >= MLP_FUSED_MIN_ROWS);
}

struct Prep {          // byte offsets into the prepared-weight buffer
  size_t wpe;
  size_t blk0, blk_stride;
  size_t wqkv, wqkvT, wproj, wprojT, wfc1, wfc1T, wfc2, wfc2T, bqkv, bfc1;   // offsets inside one block
  size_t wmlp, wmlpb;     // fc1 + fc2 as the weight streams of the fused MLP forward / backward (mlp_fused.hip)
  size_t total;
  explicit Prep(int depth) {
    size_t o = 0;
    wpe = o; o = al(o + (size_t)D * PD * 2);
    blk0 = o;
    size_t b = 0;
    wqkv = b; b = al(b + (size_t)3 * D * D * 2);
    wqkvT = b; b = al(b + (size_t)3 * D * D * 2);
    wproj = b; b = al(b + (size_t)D * D * 2);
    wprojT = b; b = al(b + (size_t)D * D * 2);
    wfc1 = b; b = al(b + (size_t)MLP * D * 2);
    wfc1T = b; b = al(b + (size_t)MLP * D * 2);
    wfc2 = b; b = al(b + (size_t)MLP * D * 2);
    wfc2T = b; b = al(b + (size_t)MLP * D * 2);
    bqkv = b; b = al(b + (size_t)3 * D * 4);
    bfc1 = b; b = al(b + (size_t)MLP * 4);
    wmlp = b; b = al(b + rovit_mlp_stream_bytes());
    wmlpb = b; b = al(b + rovit_mlp_stream_bytes());
    blk_stride = b;
    total = blk0 + (size_t)depth * blk_stride;
  }
};

struct Plan {          // byte offsets into the workspace
  int B, depth, training;
  size_t M;
  size_t X, xhat_cls, rstd_cls;
  size_t blk0, blk_stride;
  size_t xhat1, rstd1, qkv, lse, o, xhat2, rstd2, act, dact;   // inside one block
  // backward temporaries
  size_t dX, x0[3], x1[2], dpre[2], dxhat, dO, dqkv[2];     // rotating buffers of the two-stream backward (see rovit_vit_backward)
  size_t slab_qkv, slab_proj, slab_fc1, slab_fc2, slab_pe, gscr, gscr2;
  int s_qkv, s_proj, s_fc1, s_fc2, s_pe;
  int s_projc, s_fc1c, s_fc2c;      // split counts when only the B CLS rows are processed (last block)
  size_t total;
  Plan(int batch, int depth_, int training_) : B(batch), depth(depth_), training(training_) {
    M = (size_t)B * T;
    size_t o = 0;
    X = o; o = al(o + M * D * 4);
    xhat_cls = o; o = al(o + (size_t)B * D * 4);
    rstd_cls = o; o = al(o + (size_t)B * 4);
    size_t b = 0;
    xhat1 = b; b = al(b + M * D * 2);
    rstd1 = b; b = al(b + M * 4);
    qkv = b; b = al(b + M * 3 * D * 2);
    lse = b; b = al(b + (size_t)B * H * T * 4);
    this->o = b; b = al(b + M * D * 2);
    xhat2 = b; b = al(b + M * D * 2);
    rstd2 = b; b = al(b + M * 4);
    act = b; b = al(b + M * MLP * 2);
    dact = b; b = al(b + M * MLP * 2);          // gelu'(pre)
    blk_stride = training ? b : 0;              // inference: every block reuses the same buffers
    blk0 = o; o += training ? (size_t)depth * b : b;
    s_qkv = rovit_wgrad_splits((int)M, 3 * D, D);
    s_proj = rovit_wgrad_splits((int)M, D, D);
    s_fc1 = rovit_wgrad_splits((int)M, MLP, D);
    s_fc2 = rovit_wgrad_splits((int)M, D, MLP);
    s_pe = rovit_wgrad_splits(B * (T - 1), D, PD);
    s_projc = rovit_wgrad_splits(B, D, D);
    s_fc1c = rovit_wgrad_splits(B, MLP, D);
    s_fc2c = rovit_wgrad_splits(B, D, MLP);
    if (training) {
      dX = o; o = al(o + M * D * 4);
      for (int k = 0; k < 3; ++k) { x0[k] = o; o = al(o + M * D * 2); }
      for (int k = 0; k < 2; ++k) { x1[k] = o; o = al(o + M * D * 2); }
      for (int k = 0; k < 2; ++k) { dpre[k] = o; o = al(o + M * MLP * 2); }
      dxhat = o; o = al(o + M * D * 2);
      dO = o; o = al(o + M * D * 2);
      for (int k = 0; k < 2; ++k) { dqkv[k] = o; o = al(o + M * 3 * D * 2); }
      slab_qkv = o; o = al(o + rovit_wgrad_workspace_bytes(3 * D, D, s_qkv));
      slab_proj = o; o = al(o + rovit_wgrad_workspace_bytes(D, D, s_proj));
      slab_fc1 = o; o = al(o + rovit_wgrad_workspace_bytes(MLP, D, s_fc1));
      slab_fc2 = o; o = al(o + rovit_wgrad_workspace_bytes(D, MLP, s_fc2));
      slab_pe = o; o = al(o + rovit_wgrad_workspace_bytes(D, PD, s_pe));
      gscr = o; o = al(o + (size_t)MLP * D * 4);
      gscr2 = o; o = al(o + (size_t)MLP * D * 4);
    }
    total = o;
  }
};

#define RUN(call) do { int rc__ = (call); if (rc__ != ROVIT_OK) return rc__; } while (0)

// ---- second stream for the weight-gradient kernels of the backward pass -------------------------------------
// Every kernel here is a short persistent launch (20-60 us) whose ramp (dispatch, W-fragment prologue, first tile,
// tail) is ~6 us of it.  The dgrad chain (stream A = the caller's) and the wgrad/reduce kernels (stream B) of a
// block are independent given the activations, so running them on two HIP streams lets one kernel's ramp and tail
// be covered by the other's steady state.  Buffers the two streams hand over (dXb, dpre, dqkv) are double-buffered
// and every hand-over is an event; B is joined back into A before rovit_vit_backward returns.
struct SideStream {
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;
  size_t next = 0;
  // backward state carried from one block range to the next when the join was deferred (rovit_vit_backward_notify)
  hipEvent_t ev_bdone[64] = {};
  bool carry = false;
  hipEvent_t get() {
    if (next == events.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
      events.push_back(e);
    }
    return events[next++];
  }
};
SideStream* side_stream() {
  static SideStream per_dev[64];
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  SideStream& s = per_dev[d];
  if (!s.stream) {
    // A stream of its own PRIORITY class gets a hardware queue of its own.  Normal-priority streams share at most
    // GPU_MAX_HW_QUEUES (4) queues round-robin; with RCCL's and torch's side streams around, a normal-priority
    // stream was observed to land on the caller's queue, which serialises the two "concurrent" chains (8.7 instead
    // of 7.1 ms/step with a process group initialised).  High and low priority measured the same within noise: high.
    int least = 0, greatest = 0;
    const int prio = ROVIT_KNOB(ROVIT_KNOB_SIDE_PRIORITY, 0);
    if (prio == 2 || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
      if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio == 1 ? least : greatest) != hipSuccess) {
      return nullptr;
    }
  }
  return &s;
}
// record an event on `from` and (unless record_only) make `to` wait for it; returns the event (nullptr on failure).
// NB: the caller's stream is usually the NULL stream (torch's default), so a null `to` is a real stream here.
hipEvent_t hand_over(SideStream* ss, hipStream_t from, hipStream_t to, bool record_only = false) {
  hipEvent_t e = ss->get();
  if (!e || hipEventRecord(e, from) != hipSuccess) return nullptr;
  if (!record_only && hipStreamWaitEvent(to, e, 0) != hipSuccess) return nullptr;
  return e;
}
// developer library only (ROVIT_KNOB_SINGLE_STREAM): everything on the caller's stream -- serial per-kernel times for the profiles
bool two_streams_enabled() { return !ROVIT_KNOB(ROVIT_KNOB_SINGLE_STREAM, 0); }

int check_common(const void* params, const void* prep, const void* ws, int batch, int depth, int mlp_path = ROVIT_MLP_AUTO) {
  ROVIT_CHECK_ARG(params && prep && ws, ROVIT_ERR_NULL, "vit: null params/prep/workspace");
  ROVIT_CHECK_ARG(mlp_path == ROVIT_MLP_AUTO || mlp_path == ROVIT_MLP_TWO_LAUNCH || mlp_path == ROVIT_MLP_ONE_LAUNCH, ROVIT_ERR_SHAPE,
                  "vit: mlp_path must be ROVIT_MLP_AUTO, _TWO_LAUNCH or _ONE_LAUNCH (got %d)", mlp_path);
  ROVIT_CHECK_ARG(batch > 0 && depth > 0 && depth <= 64, ROVIT_ERR_SHAPE, "vit: bad batch %d / depth %d", batch, depth);
  ROVIT_CHECK_ARG(rovit_aligned16(prep) && rovit_aligned16(ws), ROVIT_ERR_ALIGN, "vit: prep/workspace must be 16-byte aligned");
  return ROVIT_OK;
}

}  // namespace

hipStream_t rovit_side_stream_handle() {
  if (!two_streams_enabled()) return nullptr;
  SideStream* s = side_stream();
  return s ? s->stream : nullptr;
}

extern "C" size_t rovit_vit_prep_bytes(int depth) { return Prep(depth).total; }
extern "C" size_t rovit_vit_workspace_bytes(int batch, int depth, int training) { return Plan(batch, depth, training).total; }
extern "C" int rovit_vit_num_params(int depth) { return P_BLOCK0 + B_COUNT * depth; }

extern "C" int rovit_vit_workspace_field(int batch, int depth, int field, int block, size_t* offset, size_t* bytes) {
  ROVIT_CHECK_ARG(offset && bytes, ROVIT_ERR_NULL, "vit_workspace_field: null output pointer");
  ROVIT_CHECK_ARG(batch > 0 && depth > 0 && block >= 0 && block < depth, ROVIT_ERR_SHAPE, "vit_workspace_field: bad batch/depth/block");
  const Plan L(batch, depth, 1);
  const size_t blk = L.blk0 + (size_t)block * L.blk_stride, M = L.M;
  switch (field) {
    case ROVIT_WS_XHAT1: *offset = blk + L.xhat1; *bytes = M * D * 2; break;
    case ROVIT_WS_RSTD1: *offset = blk + L.rstd1; *bytes = M * 4; break;
    case ROVIT_WS_QKV: *offset = blk + L.qkv; *bytes = M * 3 * D * 2; break;
    case ROVIT_WS_ATTN_O: *offset = blk + L.o; *bytes = M * D * 2; break;
    case ROVIT_WS_XHAT2: *offset = blk + L.xhat2; *bytes = M * D * 2; break;
    case ROVIT_WS_RSTD2: *offset = blk + L.rstd2; *bytes = M * 4; break;
    case ROVIT_WS_ACT: *offset = blk + L.act; *bytes = M * MLP * 2; break;
    // the last block's backward (CLS rows only behind the attention) uses buffer 0, the others their parity's
    case ROVIT_WS_DQKV: *offset = L.dqkv[block & 1]; *bytes = M * 3 * D * 2; break;
    default: rovit_set_error("vit_workspace_field: unknown field %d", field); return ROVIT_ERR_SHAPE;
  }
  return ROVIT_OK;
}

// Fold the LayerNorm affines, cast to bf16 and build the transposed copies the dgrad GEMMs read.
// Must be re-run whenever the fp32 parameters change (every optimizer step).
namespace {
// the patch-embedding weight on `s_patch` (the forward needs it first), every block's weights on `s_blocks`
int vit_prepare_impl(const float* const* params, void* prep, int depth, rovit_stream_t s_patch, rovit_stream_t s_blocks, bool gelu_tables) {
  const Prep P(depth);
  char* pb = (char*)prep;
  std::vector<RovitPrepDesc> descs;
  descs.reserve(1 + 4 * depth);
  descs.push_back({params[P_PATCH_W], nullptr, nullptr, nullptr, pb + P.wpe, nullptr, nullptr, D, PD});
  for (int i = 0; i < depth; ++i) {
    const float* const* bp = params + P_BLOCK0 + B_COUNT * i;
    char* q = pb + P.blk0 + (size_t)i * P.blk_stride;
    descs.push_back({bp[B_QKVW], bp[B_QKVB], bp[B_N1W], bp[B_N1B], q + P.wqkv, q + P.wqkvT, (float*)(q + P.bqkv), 3 * D, D});
    descs.push_back({bp[B_PROJW], nullptr, nullptr, nullptr, q + P.wproj, q + P.wprojT, nullptr, D, D});
    descs.push_back({bp[B_FC1W], bp[B_FC1B], bp[B_N2W], bp[B_N2B], q + P.wfc1, q + P.wfc1T, (float*)(q + P.bfc1), MLP, D});
    descs.push_back({bp[B_FC2W], nullptr, nullptr, nullptr, q + P.wfc2, q + P.wfc2T, nullptr, D, MLP});
  }
  if (s_patch == s_blocks) {
    RUN(rovit_prep_weight_batch(descs.data(), (int)descs.size(), s_blocks));
  } else {
    RUN(rovit_prep_weight_batch(descs.data(), 1, s_patch));
    RUN(rovit_prep_weight_batch(descs.data() + 1, (int)descs.size() - 1, s_blocks));
  }
  RUN(rovit_mlp_stream_prep_blocks(prep, P.blk0, P.blk_stride, P.wfc1, P.wfc2, P.wmlp, P.wproj, 0, P.blk_stride + P.wqkv, depth, s_blocks,
                                   gelu_tables));     // + proj and the NEXT block's qkv: the block-tail image
  RUN(rovit_mlp_stream_prep_blocks(prep, P.blk0, P.blk_stride, P.wfc2T, P.wfc1T, P.wmlpb, P.wprojT, 1, ~(size_t)0, depth, s_blocks));     // dgrad chain: (W2T, W1T) + WprojT: the backward block-tail image
  return ROVIT_OK;
}
}  // namespace

extern "C" int rovit_vit_prepare(const float* const* params, void* prep, int depth, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(params && prep && depth > 0, ROVIT_ERR_NULL, "vit_prepare: null pointer");
  return vit_prepare_impl(params, prep, depth, stream, stream, true);
}

namespace {
// prepare: 0 = the weights in `prep` are current; 1 = prepare them from `params` first; 2 = ... and write the constant tables too
int vit_forward_impl(const float* images, const float* const* params, const void* prep, void* workspace, float* features,
                     void* const* attn_taps, float* const* prob_taps, int batch, int depth, int training, int mlp_path,
                     rovit_stream_t stream, int prepare = 0) {
  ROVIT_CHECK_ARG(images && features, ROVIT_ERR_NULL, "vit_forward: null images/features");
  RUN(check_common(params, prep, workspace, batch, depth, mlp_path));
  const Prep P(depth);
  const Plan L(batch, depth, training);
  const char* pb = (const char*)prep;
  char* ws = (char*)workspace;
  float* X = (float*)(ws + L.X);
  const int M = (int)L.M;
  const float eps = 1e-6f;
  // Weight preparation inside the forward (rovit_vit_forward_prepare): four launches, 41 us per training step when they stand in front
  // of the forward.  Only the patch-embedding weight is needed at once; the blocks' weights are prepared on the side stream BESIDE the
  // patch embedding (59 us, bound by the fp32 pixels it reads), and the caller's stream waits for them in front of block 0.
  SideStream* ss_prep = (prepare && two_streams_enabled() && !attn_taps && !prob_taps && batch >= 16) ? side_stream() : nullptr;
  hipEvent_t ev_prep = nullptr;
  if (prepare) {
    if (ss_prep) {
      ss_prep->next = 0;
      if (!hand_over(ss_prep, (hipStream_t)stream, ss_prep->stream)) { rovit_set_error("vit_forward: event hand-over failed"); return ROVIT_ERR_LAUNCH; }
    }
    RUN(vit_prepare_impl(params, const_cast<void*>(prep), depth, stream, ss_prep ? (rovit_stream_t)ss_prep->stream : stream, prepare == 2));
    if (ss_prep && !(ev_prep = hand_over(ss_prep, ss_prep->stream, nullptr, true))) { rovit_set_error("vit_forward: event record failed"); return ROVIT_ERR_LAUNCH; }
  }
  RUN(rovit_cls_rows(params[P_CLS], params[P_POS], X, batch, T, stream));
  // PatchEmbed: the GEMM gathers its A tiles from the images (no im2col buffer: 77 MB and one 42 us launch less per step)
  RUN(rovit_patch_embed_fwd(images, pb + P.wpe, params[P_PATCH_B], params[P_POS], X, batch, T, stream));
  if (ev_prep && hipStreamWaitEvent((hipStream_t)stream, ev_prep, 0) != hipSuccess) { rovit_set_error("vit_forward: event wait failed"); return ROVIT_ERR_LAUNCH; }
  // Samples are independent in the forward pass, so the batch is cut into two halves that run the same kernel
  // chain on two HIP streams with no synchronisation until the final norm: every kernel here is a 20-50 us
  // persistent launch with ~6 us of ramp (dispatch, weight prologue, first tile, tail), which the other half's
  // kernels now cover.  Each half asks for half of the CUs (rovit_set_cu_budget) so the two chains co-reside.
  SideStream* ss = (two_streams_enabled() && !attn_taps && !prob_taps && batch >= 16) ? side_stream() : nullptr;
  struct Half { int b0, nb; rovit_stream_t st; };
  Half halves[2] = {{0, ss ? (batch + 1) / 2 : batch, stream}, {(batch + 1) / 2, ss ? batch / 2 : 0, ss ? (rovit_stream_t)ss->stream : stream}};
  const int nh = ss ? 2 : 1;
  if (ss) {
    if (!ss_prep) ss->next = 0;        // (the preparation above has taken events of this call already)
    if (!hand_over(ss, (hipStream_t)stream, ss->stream)) { rovit_set_error("vit_forward: event hand-over failed"); return ROVIT_ERR_LAUNCH; }
    rovit_set_cu_budget(128);
  }
  struct BudgetReset { bool on; ~BudgetReset() { if (on) rovit_set_cu_budget(256); } } budget_reset{ss != nullptr};
#define EACH_HALF for (int hh = 0; hh < nh; ++hh)
  const bool cls_fused = !attn_taps && !prob_taps;      // the last block's class-token rows: one launch (cls_tail.hip)
  bool qkv_done = false;              // the previous block's tail launch has already written this block's qkv projection
  for (int i = 0; i < depth; ++i) {
    const float* const* bp = params + P_BLOCK0 + B_COUNT * i;
    const char* q = pb + P.blk0 + (size_t)i * P.blk_stride;
    char* s = ws + L.blk0 + (size_t)i * L.blk_stride;
    // Only token 0 of the LAST block's output is consumed (final norm + heads), and everything after the
    // attention is row-wise: run proj / LN2 / MLP of that block on the B CLS rows only (row step T).
    const bool cls_only = (i == depth - 1);
    const int rs = cls_only ? T : 1;
    // per-half views: r = first row of the half; activations are row-major with the images contiguous
#define ROWS(ptr, width, esz) ((ptr) + (size_t)h.b0 * T * (width) * (esz))
    // LayerNorm1 of block 0 is a kernel of its own; every other LayerNorm of the loop is fused into the epilogue
    // of the GEMM that produces its input (proj -> norm2, fc2 -> next block's norm1).
    if (i == 0) EACH_HALF {
      const Half& h = halves[hh];
      RUN(rovit_layernorm_fwd(X + (size_t)h.b0 * T * D, ROWS(s + L.xhat1, D, 2), (float*)ROWS(s + L.rstd1, 1, 4), h.nb * T, D, eps, h.st));
    }
    if (!qkv_done) EACH_HALF {
      const Half& h = halves[hh];
      RUN(rovit_gemm_nt(ROWS(s + L.xhat1, D, 2), D, q + P.wqkv, D, h.nb * T, 3 * D, D, (const float*)(q + P.bqkv), EPI_BF16,
                        ROWS(s + L.qkv, 3 * D, 2), 3 * D, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, h.st));
    }
    qkv_done = false;
    EACH_HALF {
      const Half& h = halves[hh];
      // (developer knob 19: the second half-batch starts its first attention behind the first half's, so that one half's attention runs
      // beside the other half's block tail for the rest of the forward instead of both halves moving in phase)
      if (ROVIT_KNOB(ROVIT_KNOB_FWD_STAGGER, 0) && ss && i == 0 && hh == 1 && !hand_over(ss, (hipStream_t)stream, ss->stream)) {
        rovit_set_error("vit_forward: event hand-over failed");
        return ROVIT_ERR_LAUNCH;
      }
      // the last block: only the class token's attention output is consumed (the half behind it runs on those rows alone), and a query's
      // output needs no other query -- 197 scores per (image, head) instead of 197 x 197 (taps want every token's output: full kernel)
      if (cls_only && !attn_taps && !prob_taps) {
        RUN(rovit_attention_cls_fwd(ROWS(s + L.qkv, 3 * D, 2), ROWS(s + L.o, D, 2), (float*)(s + L.lse) + (size_t)h.b0 * H * T, h.nb, T, H, D / H,
                                    0.125f, h.st));
      } else {
        RUN(rovit_attention_fwd(ROWS(s + L.qkv, 3 * D, 2), ROWS(s + L.o, D, 2), (float*)(s + L.lse) + (size_t)h.b0 * H * T, h.nb, T, H, D / H,
                                0.125f, h.st));
      }
    }
    // explainability tap: the attention module's output (proj(attention) + bias, before the residual add) for
    // every token of block i -- what a forward hook on `blocks[i].attn` sees (reference models/backbone.py:37-62)
    if (attn_taps && attn_taps[i])
      RUN(rovit_gemm_nt(s + L.o, D, q + P.wproj, D, M, D, D, bp[B_PROJB], EPI_BF16, attn_taps[i], D, nullptr, nullptr, 0, nullptr, 0,
                        nullptr, 0, stream));
    // ... and, separately, the softmax probabilities (B,3,197,197) the reference's rollout code means to collect
    // (explainability/attention_maps.py:18-105)
    if (prob_taps && prob_taps[i]) RUN(rovit_attention_probs(s + L.qkv, prob_taps[i], batch, T, H, D / H, 0.125f, stream));
    // The last block: its post-attention half AND the final norm on the class-token rows in ONE launch (cls_tail.hip; six launches before).
    // (Taps want the attention module's output for every token: the launch-by-launch path below.)
    if (cls_only && cls_fused) {
      EACH_HALF {
        const Half& h = halves[hh];
        RUN(rovit_cls_tail_fwd(ROWS(s + L.o, D, 2), X + (size_t)h.b0 * T * D, q + P.wproj, bp[B_PROJB], q + P.wfc1, (const float*)(q + P.bfc1),
                               q + P.wfc2, bp[B_FC2B], params[P_NORM_W], params[P_NORM_B], training ? ROWS(s + L.xhat2, D, 2) : nullptr,
                               training ? (float*)ROWS(s + L.rstd2, 1, 4) : nullptr, training ? ROWS(s + L.act, MLP, 2) : nullptr,
                               training ? ROWS(s + L.dact, MLP, 2) : nullptr, features + (size_t)h.b0 * D,
                               (float*)(ws + L.xhat_cls) + (size_t)h.b0 * D, (float*)(ws + L.rstd_cls) + h.b0, h.nb, T, eps, h.st));
      }
      continue;
    }
    // Everything behind the attention in ONE launch ("block tail", mlp_fused.hip: proj + residual + norm2 + MLP + residual + next
    // norm1 + the NEXT block's qkv projection; the residual stream stays in registers between the halves).
    if (!cls_only && mlp_one_launch(mlp_path, (long)batch * T)) {
      char* sn = ws + L.blk0 + (size_t)(i + 1) * L.blk_stride;            // next block's saved-activation area
      const bool tail_qkv = true;
      qkv_done = tail_qkv;                                                // block i + 1 finds its qkv projection written
      EACH_HALF {
        const Half& h = halves[hh];
        float* Xh = X + (size_t)h.b0 * T * D;
        RUN(rovit_block_tail_fwd(ROWS(s + L.o, D, 2), q + P.wmlp, bp[B_PROJB], (const float*)(q + P.bfc1), bp[B_FC2B], Xh,
                                 training ? ROWS(s + L.xhat2, D, 2) : nullptr, training ? (float*)ROWS(s + L.rstd2, 1, 4) : nullptr,
                                 training ? ROWS(s + L.act, 32, 2) : nullptr, training ? ROWS(s + L.dact, 32, 2) : nullptr,
                                 ROWS(sn + L.xhat1, D, 2), (float*)ROWS(sn + L.rstd1, 1, 4),
                                 tail_qkv ? (const float*)(q + P.blk_stride + P.bqkv) : nullptr, tail_qkv ? ROWS(sn + L.qkv, 3 * D, 2) : nullptr,
                                 eps, h.nb * T, batch * T, h.st));
      }
      continue;
    }
    EACH_HALF {
      const Half& h = halves[hh];
      float* Xh = X + (size_t)h.b0 * T * D;
      if (cls_only) {
        RUN(rovit_gemm_nt(ROWS(s + L.o, D, 2), D * rs, q + P.wproj, D, h.nb, D, D, bp[B_PROJB], EPI_RESID, nullptr, 0, nullptr, Xh, D * rs,
                          nullptr, 0, nullptr, 0, h.st));
        RUN(rovit_layernorm_fwd_rows(Xh, ROWS(s + L.xhat2, D, 2), (float*)ROWS(s + L.rstd2, 1, 4), h.nb, T, eps, h.st));
      } else {
        RUN(rovit_gemm_resid_ln(ROWS(s + L.o, D, 2), D, q + P.wproj, D, h.nb * T, D, bp[B_PROJB], Xh, ROWS(s + L.xhat2, D, 2),
                                (float*)ROWS(s + L.rstd2, 1, 4), eps, h.st));
      }
    }
    // two-launch MLP half (small batches, and the CLS rows of the last block): fc1 + GELU, then fc2 + residual + next LayerNorm
    EACH_HALF {
      const Half& h = halves[hh];
      RUN(rovit_gemm_nt(ROWS(s + L.xhat2, D, 2), D * rs, q + P.wfc1, D, cls_only ? h.nb : h.nb * T, MLP, D, (const float*)(q + P.bfc1),
                        EPI_GELU, ROWS(s + L.act, MLP, 2), MLP * rs, training ? ROWS(s + L.dact, MLP, 2) : nullptr, nullptr, 0, nullptr, 0,
                        nullptr, 0, h.st));
    }
    EACH_HALF {
      const Half& h = halves[hh];
      float* Xh = X + (size_t)h.b0 * T * D;
      if (cls_only) {
        RUN(rovit_gemm_nt(ROWS(s + L.act, MLP, 2), MLP * rs, q + P.wfc2, MLP, h.nb, D, MLP, bp[B_FC2B], EPI_RESID, nullptr, 0, nullptr, Xh,
                          D * rs, nullptr, 0, nullptr, 0, h.st));
      } else {
        char* sn = ws + L.blk0 + (size_t)(i + 1) * L.blk_stride;          // next block's saved-activation area
        RUN(rovit_gemm_resid_ln(ROWS(s + L.act, MLP, 2), MLP, q + P.wfc2, MLP, h.nb * T, MLP, bp[B_FC2B], Xh, ROWS(sn + L.xhat1, D, 2),
                                (float*)ROWS(sn + L.rstd1, 1, 4), eps, h.st));
      }
    }
#undef ROWS
  }
#undef EACH_HALF
  if (ss && !hand_over(ss, ss->stream, (hipStream_t)stream)) { rovit_set_error("vit_forward: event hand-over failed"); return ROVIT_ERR_LAUNCH; }
  if (!cls_fused)
    RUN(rovit_cls_norm_fwd(X, params[P_NORM_W], params[P_NORM_B], features, (float*)(ws + L.xhat_cls), (float*)(ws + L.rstd_cls), batch,
                           T, eps, stream));
  return ROVIT_OK;
}
}  // namespace

// images fp32 NCHW (B,3,224,224) -> features fp32 (B,192)
extern "C" int rovit_vit_forward(const float* images, const float* const* params, const void* prep, void* workspace,
                                 float* features, int batch, int depth, int training, int mlp_path, rovit_stream_t stream) {
  return vit_forward_impl(images, params, prep, workspace, features, nullptr, nullptr, batch, depth, training, mlp_path, stream);
}

// rovit_vit_prepare + rovit_vit_forward as ONE call (a training step prepares the weights after every optimizer step): the blocks'
// weight images are written beside the patch embedding instead of in front of the forward.  write_tables != 0: also (re)write the
// constant look-up tables of `prep` (needed once per buffer; rovit_vit_prepare always writes them).
extern "C" int rovit_vit_forward_prepare(const float* images, const float* const* params, void* prep, void* workspace, float* features,
                                         int batch, int depth, int training, int mlp_path, int write_tables, rovit_stream_t stream) {
  return vit_forward_impl(images, params, prep, workspace, features, nullptr, nullptr, batch, depth, training, mlp_path, stream,
                          write_tables ? 2 : 1);
}

// Same forward (inference workspace), additionally writing each block's attention-module output into
// attn_taps[i] (bf16, (B*197,192)) and/or its softmax probabilities into prob_taps[i] (fp32, (B,3,197,197)); either
// array may be NULL, NULL entries are skipped.
extern "C" int rovit_vit_forward_taps(const float* images, const float* const* params, const void* prep, void* workspace,
                                      float* features, void* const* attn_taps, float* const* prob_taps, int batch, int depth,
                                      rovit_stream_t stream) {
  ROVIT_CHECK_ARG(attn_taps || prob_taps, ROVIT_ERR_NULL, "vit_forward_taps: no tap array given");
  return vit_forward_impl(images, params, prep, workspace, features, attn_taps, prob_taps, batch, depth, 0, ROVIT_MLP_AUTO, stream);
}

// Backward over blocks first_block, first_block-1, ..., last_block (inclusive).  first_block == depth-1 also
// runs the final-norm backward from d_features; last_block == 0 also produces the patch-embed / pos / cls
// gradients.  Splitting the range lets the caller start a gradient all-reduce between calls.
// grads[] mirrors params[]; every entry of the processed range is overwritten.
namespace {
int vit_backward_impl(const float* images, const float* d_features, const float* const* params, const void* prep, void* workspace,
                      float* const* grads, int batch, int depth, int first_block, int last_block, int mlp_path, rovit_stream_t stream,
                      bool defer_join, hipStream_t notify) {
  RUN(check_common(params, prep, workspace, batch, depth, mlp_path));
  ROVIT_CHECK_ARG(grads, ROVIT_ERR_NULL, "vit_backward: null grads");
  ROVIT_CHECK_ARG(first_block < depth && last_block >= 0 && first_block >= last_block, ROVIT_ERR_SHAPE,
                  "vit_backward: bad block range [%d..%d] for depth %d", first_block, last_block, depth);
  const Prep P(depth);
  const Plan L(batch, depth, 1);
  const char* pb = (const char*)prep;
  char* ws = (char*)workspace;
  const int M = (int)L.M;
  const bool one_launch = mlp_one_launch(mlp_path, M);       // the SAME decision the forward took (same argument, same row count)
  // (L.dX: the fp32 gradient rows of rounds 1-3; the last block's class-token chain keeps its fp32 sums on chip since cls_tail.hip)
  // Round 4: the residual-stream gradient travels between the kernels in BF16 -- every LayerNorm-backward epilogue reads the incoming
  // gradient's bf16 rows, adds its term in fp32 and writes bf16 rows; rounds 1-3 also read and wrote an fp32 dX per kernel (116 MB per
  // block at batch 256 = 11 % of the backward's bytes; bound measured first, developer knob 15: 4.465 -> 4.33 ms with the fp32 stores
  // alone skipped).  25 roundings to bf16 along the depth instead of fresh ones at every use: parameter gradients vs the fp32 oracle
  // DESIGN.md section 2.
  // The gradient ENTERING block i lives in x0[i % 3] (x0v(-1) feeds the patch embedding), the mid-block one in x1[i & 1].
  auto x0v = [&](int i) { return ws + L.x0[(i + 3) % 3]; };
  if (first_block == depth - 1) {
    ROVIT_CHECK_ARG(d_features, ROVIT_ERR_NULL, "vit_backward: null d_features");
    // (the final norm's backward opens the last block's fused class-token chain below: cls_tail.hip)
  }
  // Two-stream schedule (see SideStream above).  Per block i (p = i & 1), stream A runs the dgrad chain
  //   A1 + A2 MLP half (fc2 dgrad * gelu' -> dpre[p]; fc1 dgrad + norm2 bwd -> dX, x1[p]): one launch from batch 173, else two
  //   A3 proj dgrad : x1[p] -> dO                       A4 attention bwd : dO -> dqkv[p]
  //   A5 qkv dgrad + norm1 bwd : dqkv[p] -> dX, x0[(i-1)%3]
  // and stream B the weight gradients: ONE launch per block carries fc2 (x0[i%3], act), fc1 (dpre[p], xhat2), proj (x1[p], o) of block i
  // together with the qkv gradient (dqkv, xhat1) of the previous block (`pending`, whose dqkv became final with its A5) -- 12 output
  // tiles (192 x 192) per M-split, so 16 splits fill the chip (gemm.hip WgradProb) -- and ONE reduce launch finishes those four.
  // Events cost ~6 us of idle time on the stream that records or waits, so A does ONE record per block (E_i, after
  // A2) and ONE wait (for the reduce of block i+2, long finished).  The rotating buffers make that safe: a buffer written by A in
  // block i was last read by B in block i+2 (x1, dpre, dqkv: parity pairs) or i+3 (x0: three buffers, A5 of block i overwrites what
  // the fc2 weight gradient of block i+2 read).
  hipStream_t sA = (hipStream_t)stream;
  SideStream* ss = two_streams_enabled() ? side_stream() : nullptr;
  hipStream_t sB = ss ? ss->stream : sA;
  static hipEvent_t no_events[64];
  if (ss && !(ss->carry && first_block != depth - 1)) {      // a new backward pass, or the previous range was joined
    ss->next = 0;
    for (auto& e : ss->ev_bdone) e = nullptr;
  }
  if (ss) ss->carry = false;
  hipEvent_t* ev_bdone = ss ? ss->ev_bdone : no_events;
  int pending = -1;                                   // block whose qkv weight gradient has not been issued yet
#define EVFAIL(what) do { rovit_set_error("vit_backward: " what " failed"); return ROVIT_ERR_LAUNCH; } while (0)
  // never more splits than the slab buffers were sized for (small batches have few 64-row steps)
  const int S_MERGE = std::min(std::min(ROVIT_KNOB(ROVIT_KNOB_WGRAD_SPLITS, 16), L.s_fc1), std::min(std::min(L.s_fc2, L.s_qkv), L.s_proj));
  ROVIT_CHECK_ARG(S_MERGE >= 1, ROVIT_ERR_SHAPE, "vit_backward: no weight-gradient split fits batch %d", batch);
  auto qkv_reduce_desc = [&](int i, int splits) -> RovitReduceDesc {
    const float* const* bp = params + P_BLOCK0 + B_COUNT * i;
    float* const* bg = grads + P_BLOCK0 + B_COUNT * i;
    return {(const float*)(ws + L.slab_qkv), splits, 3 * D, D, bp[B_N1W], bp[B_N1B], bp[B_QKVW], bg[B_QKVW], bg[B_QKVB], bg[B_N1W],
            bg[B_N1B], (float*)(ws + L.gscr2)};
  };
  // the qkv weight gradient of block i as a launch of its own (flush at the end of a block range).  Same kernel, tile and split count as
  // inside a merged launch, so a backward cut into block ranges (data-parallel buckets) stays bit-identical to an uncut one.
  auto issue_qkv = [&](int i) -> int {
    char* s = ws + L.blk0 + (size_t)i * L.blk_stride;
    const RovitWgradDesc wd = {ws + L.dqkv[i & 1], 3 * D, s + L.xhat1, D, 3 * D, D, (float*)(ws + L.slab_qkv)};
    RUN(rovit_wgrad_batch(&wd, 1, M, S_MERGE, sB));
    const RovitReduceDesc rd = qkv_reduce_desc(i, S_MERGE);
    RUN(rovit_wgrad_reduce_batch(&rd, 1, sB));
    if (ss && !(ev_bdone[i] = hand_over(ss, sB, nullptr, true))) EVFAIL("event record");
    return ROVIT_OK;
  };
  // merged B-stream work of one iteration: [qkv wgrad of `prev`] + fc2, fc1, proj wgrad of block i, then one reduce
  auto issue_merged = [&](int i, int prev, const char* xin, const char* dp, const char* xmid) -> int {
    char* s = ws + L.blk0 + (size_t)i * L.blk_stride;
    const float* const* bp = params + P_BLOCK0 + B_COUNT * i;
    float* const* bg = grads + P_BLOCK0 + B_COUNT * i;
    RovitWgradDesc wd[4];
    RovitReduceDesc rd[4];
    int n = 0;
    const int blk = one_launch ? 1 : 0;                 // act and dpre chunk-major (written by the one-launch MLP kernels)
    wd[n] = {xin, D, s + L.act, MLP, D, MLP, (float*)(ws + L.slab_fc2), blk, 0};
    rd[n++] = {(const float*)(ws + L.slab_fc2), S_MERGE, D, MLP, nullptr, nullptr, nullptr, bg[B_FC2W], bg[B_FC2B], nullptr, nullptr, nullptr};
    wd[n] = {dp, MLP, s + L.xhat2, D, MLP, D, (float*)(ws + L.slab_fc1), 0, blk};
    rd[n++] = {(const float*)(ws + L.slab_fc1), S_MERGE, MLP, D, bp[B_N2W], bp[B_N2B], bp[B_FC1W], bg[B_FC1W], bg[B_FC1B], bg[B_N2W], bg[B_N2B],
               (float*)(ws + L.gscr)};
    wd[n] = {xmid, D, s + L.o, D, D, D, (float*)(ws + L.slab_proj)};
    rd[n++] = {(const float*)(ws + L.slab_proj), S_MERGE, D, D, nullptr, nullptr, nullptr, bg[B_PROJW], bg[B_PROJB], nullptr, nullptr, nullptr};
    if (prev >= 0) {
      char* sp = ws + L.blk0 + (size_t)prev * L.blk_stride;
      wd[n] = {ws + L.dqkv[prev & 1], 3 * D, sp + L.xhat1, D, 3 * D, D, (float*)(ws + L.slab_qkv)};
      rd[n++] = qkv_reduce_desc(prev, S_MERGE);
    }
    RUN(rovit_wgrad_batch(wd, n, M, S_MERGE, sB));
    RUN(rovit_wgrad_reduce_batch(rd, n, sB));
    if (prev >= 0 && ss && !(ev_bdone[prev] = hand_over(ss, sB, nullptr, true))) EVFAIL("event record");
    return ROVIT_OK;
  };
  for (int i = first_block; i >= last_block; --i) {
    const char* q = pb + P.blk0 + (size_t)i * P.blk_stride;
    char* s = ws + L.blk0 + (size_t)i * L.blk_stride;
    char* xin = x0v(i);
    char* xout = x0v(i - 1);
    // the last block's post-attention half only ever sees gradient on the B CLS rows (see rovit_vit_forward)
    if (i == depth - 1) {
      // small and serial on stream A (row step T: the CLS rows of the dense buffers).  Its three CLS-row weight gradients (256 rows
      // each: 18 us apiece as separate launches, plus their reduce) run as ONE merged launch on the weight-gradient stream: for that the
      // norm2 backward writes the bf16 gradient to the mid-block buffer instead of updating xin in place (fc2's weight gradient reads
      // xin as it came in, proj's the updated rows).
      const int Mr = batch, rs = T;
      char* dp = ws + L.dpre[i & 1];
      char* dq = ws + L.dqkv[i & 1];
      char* xmc = ws + L.x1[i & 1];
      // final-norm backward -> fc2 dgrad x gelu' -> fc1 dgrad -> norm2 backward -> proj dgrad on the class-token rows: ONE launch (five before)
      RUN(rovit_cls_tail_bwd(d_features, (const float*)(ws + L.xhat_cls), (const float*)(ws + L.rstd_cls), params[P_NORM_W], q + P.wfc2, q + P.wfc1,
                             q + P.wproj, s + L.dact, s + L.xhat2, (const float*)(s + L.rstd2), xin, dp, xmc, ws + L.dO, batch, T, stream));
      // (no zero fills, round 4: only the CLS rows of dO and of the mid-block gradient carry gradient; the attention backward and the
      // qkv dgrad below are told so and treat the other rows as zeros without reading them)
      if (ss && !hand_over(ss, sA, sB)) EVFAIL("event hand-over");
      // the final norm's dgamma / dbeta: sample sums off the dgrad chain -> the weight-gradient stream
      RUN(rovit_cls_norm_affine_grad(d_features, (const float*)(ws + L.xhat_cls), grads[P_NORM_W], grads[P_NORM_B], batch, sB));
      {
        const float* const* bpp = params + P_BLOCK0 + B_COUNT * i;
        float* const* bg = grads + P_BLOCK0 + B_COUNT * i;
        const int sc = std::min(std::min(4, L.s_fc2c), std::min(L.s_fc1c, L.s_projc));
        const RovitWgradDesc wd[3] = {{xin, D * rs, s + L.act, MLP * rs, D, MLP, (float*)(ws + L.slab_fc2)},
                                      {dp, MLP * rs, s + L.xhat2, D * rs, MLP, D, (float*)(ws + L.slab_fc1)},
                                      {xmc, D * rs, s + L.o, D * rs, D, D, (float*)(ws + L.slab_proj)}};
        RUN(rovit_wgrad_batch(wd, 3, Mr, sc, sB));
        const RovitReduceDesc rd[3] = {
            {(const float*)(ws + L.slab_fc2), sc, D, MLP, nullptr, nullptr, nullptr, bg[B_FC2W], bg[B_FC2B], nullptr, nullptr, nullptr},
            {(const float*)(ws + L.slab_fc1), sc, MLP, D, bpp[B_N2W], bpp[B_N2B], bpp[B_FC1W], bg[B_FC1W], bg[B_FC1B], bg[B_N2W], bg[B_N2B],
             (float*)(ws + L.gscr)},
            {(const float*)(ws + L.slab_proj), sc, D, D, nullptr, nullptr, nullptr, bg[B_PROJW], bg[B_PROJB], nullptr, nullptr, nullptr}};
        RUN(rovit_wgrad_reduce_batch(rd, 3, sB));
      }
      // only the class token's row of dO carries gradient: a rank-one backward (attention.hip), which also zeroes the other queries' dQ rows
      RUN(rovit_attention_cls_bwd(s + L.qkv, s + L.o, (const float*)(s + L.lse), ws + L.dO, dq, batch, T, H, D / H, 0.125f, stream));
      RUN(rovit_gemm_ln_bwd_cls(dq, 3 * D, q + P.wqkvT, 3 * D, M, 3 * D, s + L.xhat1, (const float*)(s + L.rstd1), xmc, T, xout, stream));
      // the (full-size) qkv weight gradient of this block goes the way of every other block's: as `pending`, into the next block's
      // merged launch on the weight-gradient stream (or the flush behind the loop) instead of 50 us of serial work here
      pending = i;
      continue;
    }
    const int p = i & 1;
    char* dp = ws + L.dpre[p];
    char* dq = ws + L.dqkv[p];
    char* xmid = ws + L.x1[p];
    if (ss && i + 2 < depth && ev_bdone[i + 2] && hipStreamWaitEvent(sA, ev_bdone[i + 2], 0) != hipSuccess) EVFAIL("event wait");
    // A1 + A2 in one launch (mlp_fused.hip): fc2 dgrad x gelu' -> dpre (kept for the fc1 weight gradient), fc1 dgrad + norm2 backward
    // without re-reading dpre; two launches below the batch threshold.  (Round 3 also built the backward's counterpart of the block tail
    // -- norm2 backward in registers and the proj dgrad in the same launch -- and measured it SLOWER in the step, 4.87 against 4.78 ms:
    // tools/attic)
    if (one_launch) {
      RUN(rovit_mlp_fused_bwd(xin, q + P.wmlpb, s + L.dact, dp, s + L.xhat2, (const float*)(s + L.rstd2), nullptr, xmid, M, sA));
    } else {
      RUN(rovit_gemm_nt(xin, D, q + P.wfc2T, D, M, MLP, D, nullptr, EPI_MUL, dp, MLP, nullptr, nullptr, 0, s + L.dact, MLP, nullptr, 0, sA));   // A1
      // fc1 dgrad fused with the backward of norm2 (updates dX, writes its bf16 copy)
      RUN(rovit_gemm_ln_bwd(dp, MLP, q + P.wfc1T, MLP, M, MLP, s + L.xhat2, (const float*)(s + L.rstd2), nullptr, xin, xmid, sA));   // A2
    }
    if (ss && !hand_over(ss, sA, sB)) EVFAIL("event hand-over");                                                   // E_i
    RUN(issue_merged(i, pending, xin, dp, xmid));
    {
      const int cus = ROVIT_KNOB(ROVIT_KNOB_PROJ_DGRAD_CUS, 256);
      if (cus != 256) rovit_set_cu_budget(cus);
      const int rc_a3 = rovit_gemm_nt(xmid, D, q + P.wprojT, D, M, D, D, nullptr, EPI_BF16, ws + L.dO, D, nullptr, nullptr, 0, nullptr, 0, nullptr, 0,
                                      sA);                                                                         // A3
      if (cus != 256) rovit_set_cu_budget(256);
      if (rc_a3 != ROVIT_OK) return rc_a3;
    }
    RUN(rovit_attention_bwd(s + L.qkv, s + L.o, (const float*)(s + L.lse), ws + L.dO, dq, batch, T, H, D / H, 0.125f, sA));       // A4
    // qkv dgrad fused with the backward of norm1
    RUN(rovit_gemm_ln_bwd(dq, 3 * D, q + P.wqkvT, 3 * D, M, 3 * D, s + L.xhat1, (const float*)(s + L.rstd1), nullptr, xmid, xout, sA));   // A5
    pending = i;
  }
  auto patch_grads = [&]() -> int {          // the patch embedding's and the position embedding's gradients (block range ending at 0)
    ROVIT_CHECK_ARG(images, ROVIT_ERR_NULL, "vit_backward: the range ending at block 0 needs the images of the forward call");
    RUN(rovit_patch_embed_wgrad(x0v(-1), D, images, batch, T, D, L.s_pe, (float*)(ws + L.slab_pe), stream));
    RUN(rovit_wgrad_reduce((const float*)(ws + L.slab_pe), L.s_pe, D, PD, nullptr, nullptr, nullptr, grads[P_PATCH_W], grads[P_PATCH_B],
                           nullptr, nullptr, nullptr, stream));
    RUN(rovit_pos_grad(nullptr, x0v(-1), grads[P_POS], grads[P_CLS], batch, T, stream));
    return ROVIT_OK;
  };
  bool patch_done = false;
  if (pending >= 0) {
    if (ss && !hand_over(ss, sA, sB)) EVFAIL("event hand-over");
    RUN(issue_qkv(pending));
    // the patch-embedding weight gradient (58 us, needs only the dgrad chain's final dX) runs on the caller's stream BESIDE block 0's
    // last weight gradients on the side stream instead of behind the join
    if (ss && last_block == 0) { RUN(patch_grads()); patch_done = true; }
    if (ss && defer_join && last_block > 0) {
      // the gradients of this range are final once B drains: tell the caller's reduction stream, do not stall A
      if (!hand_over(ss, sB, notify)) EVFAIL("event hand-over");
      ss->carry = true;
    } else if (ss && !hand_over(ss, sB, sA)) {       // join: the gradients of the range are final on A
      EVFAIL("event hand-over");
    }
  }
#undef EVFAIL
  if (last_block == 0 && !patch_done) RUN(patch_grads());
  return ROVIT_OK;
}
}  // namespace

extern "C" int rovit_vit_backward(const float* images, const float* d_features, const float* const* params, const void* prep,
                                  void* workspace, float* const* grads, int batch, int depth, int first_block, int last_block,
                                  int mlp_path, rovit_stream_t stream) {
  return vit_backward_impl(images, d_features, params, prep, workspace, grads, batch, depth, first_block, last_block, mlp_path, stream, false,
                           nullptr);
}

// Same, for a data-parallel caller that reduces each block range while the next one runs: for last_block > 0 the
// weight-gradient stream is NOT joined back into `stream`; instead `notify_stream` (the caller's reduction stream) is
// made to wait until the gradients of this range are final.  Ranges must then be issued in order down to
// last_block == 0, whose call joins everything into `stream`.
extern "C" int rovit_vit_backward_notify(const float* images, const float* d_features, const float* const* params, const void* prep,
                                         void* workspace, float* const* grads, int batch, int depth, int first_block,
                                         int last_block, int mlp_path, rovit_stream_t stream, rovit_stream_t notify_stream) {
  return vit_backward_impl(images, d_features, params, prep, workspace, grads, batch, depth, first_block, last_block, mlp_path, stream, true,
                           (hipStream_t)notify_stream);
}
