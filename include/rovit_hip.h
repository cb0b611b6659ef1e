/* rovit_hip.h -- C ABI of librovit_hip.so: the MI355X (gfx950) kernels behind the RoViT-KAN hot path.
 *
 * The reference (nishitbohra/RoViT-KAN-...) has no native / FFI layer: its hot path is the Python nn.Module
 * surface of models/rovit_kan.py, models/backbone.py, models/kan.py and models/heads.py.  This header is the
 * boundary a maintainer would bind from those files (via ctypes); every entry point cites
 * the reference code it replaces.
 *
 * Conventions
 *   - plain C types only; all pointers are DEVICE pointers unless stated, owned by the caller;
 *   - "bf16" buffers are passed as void* (16-bit brain-float, row-major, 16-byte aligned, leading dimension in
 *     elements); float buffers are fp32 row-major;
 *   - every function only ENQUEUES work on `stream` (a hipStream_t); it never allocates, never synchronises;
 *   - return value: ROVIT_OK (0) or a negative error code; rovit_last_error_string() describes the failure
 *     (thread-local).  There is no CPU fallback: without a GPU the launch fails and the code says so.
 */
#ifndef ROVIT_HIP_H
#define ROVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rovit_stream_t; /* hipStream_t */

enum {
  ROVIT_OK = 0,
  ROVIT_ERR_SHAPE = -1,  /* unsupported / inconsistent shape */
  ROVIT_ERR_ALIGN = -2,  /* pointer or leading dimension not aligned */
  ROVIT_ERR_NULL = -3,   /* required pointer is NULL */
  ROVIT_ERR_LAUNCH = -4  /* HIP reported a launch / memset error */
};

/* activation applied to a KAN layer's output (models/kan.py:141 ReLU between layers, :147 3*sigmoid) */
enum { ROVIT_ACT_NONE = 0, ROVIT_ACT_RELU = 1, ROVIT_ACT_SIGMOID3 = 2 };
/* flags of rovit_linear_fwd */
enum { ROVIT_LIN_RELU = 1, ROVIT_LIN_CLAMP10 = 2 };
/* epilogues of rovit_gemm_nt */
enum { ROVIT_EPI_BF16 = 0, ROVIT_EPI_GELU = 1, ROVIT_EPI_RESID = 2, ROVIT_EPI_MUL = 3, ROVIT_EPI_PATCH = 4 };
/* OR-ed into rovit_gemm_nt's `epi`: run the LDS-tiled kernel (128 x 192 / 128 x 96 tiles; the path of shapes the weight-stationary
 * kernels do not cover) even where a weight-stationary kernel applies -- per call, for tests of that path */
enum { ROVIT_GEMM_TILED_192 = 0x100, ROVIT_GEMM_TILED_96 = 0x200 };

int rovit_version(void);
const char* rovit_last_error_string(void);

/* ------------------------------------------------------------------------------------------------------------
 * KAN head.  Replaces KANLayer.forward (models/kan.py:70-95) incl. BSplineBasis.compute_basis (:8-44) and the
 * activation that follows the layer in KANSeverityModule.forward (:138-149); backward replaces autograd of same.
 *   x (B,in)  spline_w (in,out,nb)  knots (n_knots,) with nb = n_knots-4 (degree 3)  lin_w (out,in)  lin_b (out)
 *   out (B,out) = act(Linear(x) + sum_i sum_k basis_k(tanh x_i) spline_w[i,:,k])
 * ------------------------------------------------------------------------------------------------------------ */
/* BSplineBasis.compute_basis (models/kan.py:8-44) for degree 3: x_norm (n,) -> basis (n, n_knots-4) */
int rovit_kan_basis(const float* x_norm, const float* knots, float* basis, int n, int n_knots, rovit_stream_t stream);
int rovit_kan_layer_fwd(const float* x, const float* spline_w, const float* knots, const float* lin_w, const float* lin_b,
                        float* out, int batch, int in_f, int out_f, int n_knots, int act, rovit_stream_t stream);
/* out = the forward's post-activation output; dx may be NULL; d_spline_w/d_lin_w/d_lin_b NULL together */
int rovit_kan_layer_bwd(const float* x, const float* spline_w, const float* knots, const float* lin_w, const float* out,
                        const float* grad_out, float* dx, float* d_spline_w, float* d_lin_w, float* d_lin_b, int batch,
                        int in_f, int out_f, int n_knots, int act, int accumulate_dx, rovit_stream_t stream);
/* Backward of the whole KANSeverityModule stack in two launches (autograd of models/kan.py:138-149): the per-sample chain
 * dL/dz_n -> dx_n -> ... -> dx_1 in one launch (writes dL/dz of every layer into gz[l] (batch, out_l) and dx), then the
 * parameter gradients of ALL layers in one launch.  Host arrays of n_layers device pointers; grad_outs[l] = gradient w.r.t.
 * layer l's output from outside the stack (NULL entries allowed, the last must be set); d_spline_w NULL = no parameter
 * gradients; dx NULL = no input gradient.  Same arithmetic and summation order as rovit_kan_layer_bwd per layer. */
int rovit_kan_stack_bwd(const float* x, const float* const* spline_w, const float* const* knots, const float* const* lin_w,
                        const float* const* outs, const float* const* grad_outs, float* const* gz, float* dx, float* const* d_spline_w,
                        float* const* d_lin_w, float* const* d_lin_b, int batch, const int* dims, const int* n_knots, const int* acts,
                        int n_layers, rovit_stream_t stream);
/* Prepared weight layouts of one KAN layer for rovit_kan_stack_fwd (re-run whenever the parameters change):
 * spline_w (in, out, nb) -> spline_wt (in, nb, out); lin_w (out, in) -> lin_wt (in, out). */
int rovit_kan_prepare(const float* spline_w, const float* lin_w, float* spline_wt, float* lin_wt, int in_f, int out_f, int n_basis,
                      rovit_stream_t stream);
/* KANSeverityModule.forward (models/kan.py:138-149) in ONE launch: every layer with its activation; the activations
 * stay on the CU between layers.  spline_wt / knots / lin_wt / lin_b / outs are HOST arrays of n_layers device pointers
 * (spline_wt, lin_wt: the prepared layouts above), dims the n_layers + 1 widths (widths after the input <= 64), acts the
 * ROVIT_ACT_* after each layer.  outs[l] (batch, dims[l+1]) receives layer l's post-activation output: the last is the
 * module output, the others are what rovit_kan_layer_bwd and get_activation_trajectory (:154-167) need. */
int rovit_kan_stack_fwd(const float* x, const float* const* spline_wt, const float* const* knots, const float* const* lin_wt,
                        const float* const* lin_b, float* const* outs, int batch, const int* dims, const int* n_knots,
                        const int* acts, int n_layers, rovit_stream_t stream);
/* The same stack on the matrix cores for large batches (models/kan.py:70-95 as the dense contraction
 * sum_j sum_s R[b,j,s] Wd[j,s,o]: slots s = the num_basis truncated-basis values of tanh(x_j) followed by the raw x_j of
 * the layer's Linear term; v_mfma_f32_32x32x2_f32, fp32 operands and accumulation).  rovit_kan_mfma_prepared_floats gives
 * the size of one layer's prepared weight layout, or 0 when the kernel does not cover the layer (it covers in_f % 8 == 0,
 * out_f <= 64 and n_basis 7 or 34, i.e. num_knots 5 and 32 of the reference's configs); rovit_kan_prepare_mfma fills it
 * (re-run whenever the parameters change); wm / knots / lin_b / outs are HOST arrays of n_layers device pointers. */
size_t rovit_kan_mfma_prepared_floats(int in_f, int out_f, int n_basis);
int rovit_kan_prepare_mfma(const float* spline_w, const float* lin_w, float* wm, int in_f, int out_f, int n_basis, rovit_stream_t stream);
int rovit_kan_stack_fwd_mfma(const float* x, const float* const* wm, const float* const* knots, const float* const* lin_b,
                             float* const* outs, int batch, const int* dims, const int* n_knots, const int* acts, int n_layers,
                             rovit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * MLP heads.  rovit_linear_* are the building block (nn.Linear [+ReLU] [*dropout mask] [clamp +-10]);
 * rovit_heads_* run ClassificationHead / OrdinalHead / UncertaintyHead.forward (models/heads.py:17-22, 38-43,
 * 91-102) with the curriculum gate of RoViTKAN.forward (models/rovit_kan.py:93-116).
 * params[14] / grads[14] order: cls.fc1.{w,b} cls.fc2.{w,b} ord.fc1.{w,b} ord.fc2.{w,b} unc.fc1.{w,b}
 * unc.fc_mu.{w,b} unc.fc_logvar.{w,b}.  masks: HOST array of 3 device pointers (B,hid) holding the scaled
 * dropout keep-mask, or NULL / NULL entries in eval mode.  hidden: (3,B,hid) workspace kept for backward;
 * rovit_heads_bwd's scratch is (3,B,hid) as well.
 * ------------------------------------------------------------------------------------------------------------ */
int rovit_linear_fwd(const float* x, const float* w, const float* bias, const float* mask, float* y, int batch, int in_f,
                     int out_f, int flags, rovit_stream_t stream);
int rovit_linear_bwd(const float* x, const float* w, const float* grad_y, const float* y_clamped, const float* dx_mul,
                     const float* dx_pos, float* dx, float* dw, float* db, int batch, int in_f, int out_f, int accumulate_dx,
                     rovit_stream_t stream);
int rovit_heads_fwd(const float* features, const float* const* params, const float* const* masks, float* hidden,
                    float* cls_logits, float* ordinal_logits, float* mu, float* log_var, int batch, int embed, int hid,
                    int num_classes, int stage, rovit_stream_t stream);
int rovit_heads_bwd(const float* features, const float* const* params, const float* const* masks, const float* hidden,
                    const float* log_var, const float* g_cls, const float* g_ord, const float* g_mu, const float* g_lv,
                    float* d_features, float* const* grads, float* scratch, int batch, int embed, int hid, int num_classes,
                    int accumulate_dfeat, rovit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Head phase in three launches (round 4): everything RoViTKAN.forward does with the backbone features --
 * the three heads (models/heads.py:17-22, 38-43, 91-102, gated by the curriculum stage, models/rovit_kan.py:93-116)
 * AND KANSeverityModule.forward (models/kan.py:138-149) -- as ONE forward launch, and its autograd backward as
 * two (the per-sample gradient chain down to d_features; then every parameter gradient).  One workgroup owns one
 * sample for the whole phase; parameters are read in the reference layouts (no prepared copies).
 *   Limits: embed <= 768 and a multiple of 4; hid <= 256 and a multiple of 4; num_classes <= 8; 1..4 KAN layers
 *   (kan_layers == 0: no KAN stack, stage < 4) whose widths after the input are <= 64; 8..64 knots per layer.
 *   Dropout on the hidden layers: masks[h] (B,hid) = scaled keep-mask, or -- masks[h] == NULL and drop_p > 0 -- drawn
 *   in the kernel (Philox4x32-10 keyed by `seed`, counter (sample * hid + unit, offset); kept units scaled by
 *   1 / (1 - drop_p)); the backward needs no mask then (a kept unit is one whose stored hidden value is > 0).
 *   Arrays in this struct are HOST arrays of device pointers; every pointer is a device pointer of fp32 data.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct rovit_head_phase {
  int batch, embed, hid, num_classes, stage;
  int kan_layers;
  int kan_dims[5];
  int kan_knots[4];
  int kan_acts[4];                /* ROVIT_ACT_* after each layer */
  float drop_p;
  unsigned long long seed, offset;
  const float* features;          /* (B, embed) */
  const float* head_params[14];   /* order of rovit_heads_fwd */
  const float* masks[3];
  const float* kan_w[4];          /* (in, out, nb) */
  const float* kan_knots_p[4];
  const float* kan_lw[4];         /* (out, in) */
  const float* kan_lb[4];
  /* forward outputs, kept for the backward */
  float* hidden;                  /* (3, B, hid) post-ReLU / dropout */
  float* cls; float* ord; float* mu; float* lv;
  float* kan_out[4];              /* (B, kan_dims[l+1]) post-activation */
  /* backward inputs: gradients w.r.t. the outputs (NULL: none); g_kan is the last KAN layer's */
  const float* g_cls; const float* g_ord; const float* g_mu; const float* g_lv; const float* g_kan;
  /* backward outputs */
  float* d_features;              /* (B, embed) or NULL */
  float* dpre;                    /* (3, B, hid) scratch: gradient w.r.t. the heads' pre-activations */
  float* kan_gz[4];               /* (B, kan_dims[l+1]) scratch: gradient w.r.t. each layer's pre-activation */
  float* head_grads[14];          /* mirrors head_params; all NULL with want_param_grads == 0 */
  float* kan_dw[4]; float* kan_dlw[4]; float* kan_dlb[4];
  int want_param_grads;
} rovit_head_phase;
int rovit_head_phase_fwd(const rovit_head_phase* p, rovit_stream_t stream);
int rovit_head_phase_bwd(const rovit_head_phase* p, rovit_stream_t stream);
/* the parameter-gradient launch of rovit_head_phase_bwd alone (after a want_param_grads == 0 call; e.g. on another stream) */
int rovit_head_phase_bwd_params(const rovit_head_phase* p, rovit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * DeiT-Tiny backbone (models/backbone.py:23-25 -> timm VisionTransformer.forward).
 * params / grads: HOST arrays of rovit_vit_num_params(depth) device pointers (fp32, timm layouts):
 *   [0] cls_token (192) [1] pos_embed (197,192) [2] patch_embed.proj.weight (192,768) [3] .bias [4] norm.weight
 *   [5] norm.bias, then per block b at 6+12b: norm1.{w,b} attn.qkv.{w,b} attn.proj.{w,b} norm2.{w,b} mlp.fc1.{w,b}
 *   mlp.fc2.{w,b}.
 * prep: rovit_vit_prep_bytes(depth) bytes, filled by rovit_vit_prepare (re-run after every parameter update).
 * workspace: rovit_vit_workspace_bytes(batch, depth, training) bytes; in training mode it carries the saved
 * activations from rovit_vit_forward to rovit_vit_backward.
 * ------------------------------------------------------------------------------------------------------------ */
int rovit_vit_num_params(int depth);
size_t rovit_vit_prep_bytes(int depth);
size_t rovit_vit_workspace_bytes(int batch, int depth, int training);
int rovit_vit_prepare(const float* const* params, void* prep, int depth, rovit_stream_t stream);
/* mlp_path: which kernels run the MLP half of every block -- an argument, not library state, because a training step's forward and
 * backward must agree on it (the one-launch kernels keep act / gelu' / dpre chunk-major, the two-launch kernels row-major): pass the SAME
 * value to rovit_vit_forward and to the rovit_vit_backward(_notify) calls that consume its workspace.  ROVIT_MLP_AUTO (what the module
 * passes) selects by size: one launch from 34 000 token rows (batch 173), the measured crossover; the other two values force a path
 * (parity tests run every batch size through both). */
enum { ROVIT_MLP_AUTO = 0, ROVIT_MLP_TWO_LAUNCH = 1, ROVIT_MLP_ONE_LAUNCH = 2 };
int rovit_vit_forward(const float* images, const float* const* params, const void* prep, void* workspace, float* features,
                      int batch, int depth, int training, int mlp_path, rovit_stream_t stream);
/* rovit_vit_prepare + rovit_vit_forward in one call (every training step re-prepares the weights): the per-block weight images are
 * written on the library's second stream beside the patch embedding, not in front of the forward (41 us per step at depth 12).
 * write_tables != 0 also writes prep's constant look-up tables: needed the first time a prep buffer is used. */
int rovit_vit_forward_prepare(const float* images, const float* const* params, void* prep, void* workspace, float* features, int batch,
                              int depth, int training, int mlp_path, int write_tables, rovit_stream_t stream);
/* forward + explainability taps: attn_taps is a HOST array of `depth` device pointers (bf16 (B*197,192)) that receive
 * each block's attention-module output -- what DeiTTinyBackbone.get_attention_maps collects through forward hooks on
 * `blocks[i].attn` (models/backbone.py:37-62).  prob_taps (optional, like attn_taps): fp32 (B,3,197,197) softmax
 * probabilities per block, what explainability/attention_maps.py:18-105 means to roll out.  Inference workspace. */
int rovit_vit_forward_taps(const float* images, const float* const* params, const void* prep, void* workspace, float* features,
                           void* const* attn_taps, float* const* prob_taps, int batch, int depth, rovit_stream_t stream);
/* images: the batch the forward ran on (read by the patch-embedding weight gradient, which gathers its pixels from it:
 * there is no im2col buffer); may be NULL for ranges with last_block > 0. */
int rovit_vit_backward(const float* images, const float* d_features, const float* const* params, const void* prep, void* workspace,
                       float* const* grads, int batch, int depth, int first_block, int last_block, int mlp_path, rovit_stream_t stream);
/* fp32 reference-precision forward (inference only; parity / evaluation mode, not the fast path): the same arithmetic
 * with every operand, product and sum in fp32 -- the mode in which BASELINE.json's "logits/severity within 1e-3 (fp32),
 * class argmax bit-exact" is checked end to end.  params: the ORIGINAL fp32 parameters (no prepared weights);
 * workspace: rovit_vit_f32_workspace_bytes(batch) bytes.  From batch 192 up the call runs the batch as two half-batch
 * chains, one on `stream` and one on the library's side stream (forked from and joined back into `stream`: to the caller it
 * is one stream-ordered call); the result does not depend on the batch an image travels in. */
size_t rovit_vit_f32_workspace_bytes(int batch);
int rovit_vit_forward_f32(const float* images, const float* const* params, void* workspace, float* features, int batch, int depth,
                          rovit_stream_t stream);
/* Where a saved activation / backward temporary of block `block` lives inside a TRAINING workspace
 * (rovit_vit_workspace_bytes(batch, depth, 1)): byte offset in *offset, extent in *bytes.  This is what the
 * explainability taps read (reference explainability/gradcam.py:18-60 hooks blocks[-1].norm1 for activations and
 * gradients; attention_maps.py:24-32 hooks blocks[i].attn): the fused kernels' own buffers, no extra copy.
 *   XHAT1 / XHAT2: bf16 (M,192) normalised rows before the norm1 / norm2 affine; RSTD1 / RSTD2: fp32 (M)
 *   QKV: bf16 (M,576); ATTN_O: bf16 (M,192) attention output before proj; ACT: bf16 gelu(fc1), (M,768) row-major when the
 *   two-launch MLP half ran (mlp_path, see rovit_vit_forward), CHUNK-MAJOR [24][M][32] when the one-launch half did
 *   DQKV: bf16 (M,576) gradient w.r.t. the qkv output of `block`, valid after rovit_vit_backward has processed that
 *   block and before it processes block-2 (call it with first_block = last_block = block, then read) */
enum { ROVIT_WS_XHAT1 = 0, ROVIT_WS_RSTD1 = 1, ROVIT_WS_QKV = 2, ROVIT_WS_ATTN_O = 3, ROVIT_WS_XHAT2 = 4, ROVIT_WS_RSTD2 = 5,
       ROVIT_WS_ACT = 6, ROVIT_WS_DQKV = 7 };
int rovit_vit_workspace_field(int batch, int depth, int field, int block, size_t* offset, size_t* bytes);
/* rovit_vit_backward for a data-parallel caller: for last_block > 0 the call does not wait for the range's weight
 * gradients on `stream`; `notify_stream` (the caller's reduction stream) is made to wait for them instead.  Issue the
 * ranges in order down to last_block == 0; that call joins everything into `stream`. */
int rovit_vit_backward_notify(const float* images, const float* d_features, const float* const* params, const void* prep,
                              void* workspace, float* const* grads, int batch, int depth, int first_block, int last_block,
                              int mlp_path, rovit_stream_t stream, rovit_stream_t notify_stream);

/* ---- the individual backbone kernels (used by rovit_vit_* and exposed for unit tests / profiling) ---------- */
/* C = A(M,K) W(N,K)^T + bias with a fused epilogue:
 *   BF16  out bf16 (M,N)                      GELU  out = gelu(c), out2 = gelu'(c)  (both bf16, ld = ldo)
 *   RESID xres fp32 (M,N) += c                MUL   out = c * mul (bf16)
 *   PATCH row m=(b,p) of M=B*(tokens-1) -> xres[b*tokens+1+p] = c + pos[1+p]                                 */
int rovit_gemm_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, int epi, void* out,
                  int ldo, void* out2, float* xres, int ldx, const void* mul, int ldm, const float* pos, int tokens,
                  rovit_stream_t stream);
/* X(M,192) += bf16(A W^T + bias), fused with the LayerNorm that follows the residual add (timm Block: x = x + f(x);
 * norm(x)): xhat_out bf16 (M,192) and rstd_out (M) of the updated rows; xhat_out NULL = residual add only. */
int rovit_gemm_resid_ln(const void* A, int lda, const void* W, int ldw, int M, int K, const float* bias, float* X, void* xhat_out,
                        float* rstd_out, float eps, rovit_stream_t stream);
/* The MLP half of a block in ONE launch (round 3; timm Block: x = x + mlp(norm2(x)), then the next norm1 -- timm `Mlp` and
 * `Block` reached through models/backbone.py:23-25):  X (M,192) += fc2(GELU(fc1(xhat2))) + biases, xhat_out / rstd_out = the
 * LayerNorm of the updated rows (xhat_out NULL = residual add only).  act / dact (bf16 (M,768): GELU(pre) and GELU'(pre), what
 * the backward needs) may both be NULL (inference: nothing is kept), or dact alone.  `wstream` is the weight image written
 * by rovit_mlp_prepare_stream (rovit_mlp_stream_bytes() bytes) from w1f = the bf16 fc1 weight with the norm2 affine folded in
 * (rovit_prep_weight's Wf, (768,192)) and w2 = the bf16 fc2 weight (192,768); b1 = the folded fc1 bias (768), b2 (192).
 * The VALUES of act and dact are bit-identical to rovit_gemm_nt(ROVIT_EPI_GELU)'s outputs; their LAYOUT is chunk-major: element
 * (row m, hidden unit h) of a tensor of act_rows rows at ((h / 32) * act_rows + m) * 32 + h % 32, so that a 16-row tile's store is one
 * contiguous kilobyte (row-major the launch took 94 us once its 155 MB of outputs no longer fit the Infinity Cache, chunk-major 76).
 * A launch may cover a row range of the tensors: pass act / dact advanced by first_row * 32 elements, M = rows of the range,
 * act_rows = rows of the whole tensors (= M for a whole-batch call). */
size_t rovit_mlp_stream_bytes(void);
int rovit_mlp_prepare_stream(const void* w1f, const void* w2, void* wstream, rovit_stream_t stream);
int rovit_mlp_fused_fwd(const void* xhat2, const void* wstream, const float* b1, const float* b2, void* act, void* dact, float* X,
                        void* xhat_out, float* rstd_out, float eps, int M, int act_rows, rovit_stream_t stream);
/* Everything of a block behind the attention in ONE launch (timm Block: x = x + proj(attn(norm1 x)); x = x + mlp(norm2 x); then the next
 * block's norm1 -- models/backbone.py:23-25):  X += o Wp^T + bp;  xhat2 / rstd2 = LayerNorm(X) (kept for the backward; NULL: inference);
 * X += fc2(GELU(fc1(xhat2)));  xhat_out / rstd_out = LayerNorm(X) (NULL: none).  o: bf16 (M,192) attention output; wstream from
 * rovit_mlp_prepare_stream_tail (w1f, w2 as rovit_mlp_prepare_stream; wproj = the bf16 (192,192) proj weight); act / dact chunk-major
 * as rovit_mlp_fused_fwd.  The residual stream stays in fp32 registers between the halves (nothing staged through bf16: closer to the
 * fp32 reference than proj + rovit_mlp_fused_fwd as two launches, not bit-identical to them).
 * With wqkv_next (the NEXT block's bf16 qkv weight (576,192), its norm1 affine folded in) given to the preparation and bq_next / qkv_next
 * given to the launch, the launch also writes that block's qkv projection qkv_next (M,576) = xhat_out Wqkv^T + bq_next: the forward
 * of a block is then two launches, attention and this one. */
int rovit_mlp_prepare_stream_tail(const void* w1f, const void* w2, const void* wproj, const void* wqkv_next, void* wstream,
                                  rovit_stream_t stream);
int rovit_block_tail_fwd(const void* o, const void* wstream, const float* bp, const float* b1, const float* b2, float* X, void* xhat2,
                         float* rstd2, void* act, void* dact, void* xhat_out, float* rstd_out, const float* bq_next, void* qkv_next,
                         float eps, int M, int act_rows, rovit_stream_t stream);
/* The dgrad chain of the same half in ONE launch (autograd of the above, training/trainer.py:119,136):
 *   dpre (M,768) = (dY (M,192) W2T^T) * dact        -- kept: the fc1 weight gradient reads it (bit-identical to rovit_gemm_nt(ROVIT_EPI_MUL))
 *   dX (M,192) += rstd2 (g - mean(g) - xhat2 mean(g xhat2)),  g = dpre W1T^T;   dXb = bf16(dX)     (= rovit_gemm_ln_bwd)
 * wstream_bwd: rovit_mlp_prepare_stream(w1f := W2T bf16 (768,192), w2 := W1T bf16 (192,768), norm2 affine folded in).
 * dact (input) and dpre (output) are chunk-major [24][M][32] like the forward's act / dact; rovit_wgrad_multi_ex reads them as such. */
/* dX == NULL (round 4, what rovit_vit_backward does): the residual gradient travels in bf16 -- the incoming gradient is dY itself,
 * dXb = bf16(float(dY) + the LayerNorm-backward term), and no fp32 dX is read or written. */
int rovit_mlp_fused_bwd(const void* dY, const void* wstream_bwd, const void* dact, void* dpre, const void* xhat2, const float* rstd2,
                        float* dX, void* dXb, int M, rovit_stream_t stream);
/* dgrad through a Linear that follows a LayerNorm, fused with that LayerNorm's backward:
 * dxhat = dY W^T;  dX += rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat));  dXb = bf16(dX).
 * dXb_in != NULL (round 4): the incoming residual gradient as bf16 rows (M,192); then dXb = bf16(float(dXb_in) + that term) and the fp32
 * dX (may be NULL) is neither read nor written: 58 MB less per launch at batch 256. */
int rovit_gemm_ln_bwd(const void* dY, int ldy, const void* W, int ldw, int M, int K, const void* xhat, const float* rstd, float* dX,
                      const void* dXb_in, void* dXb, rovit_stream_t stream);
/* G(N,K) = dY(M,N)^T A(M,K) and colsum(dY), split over M into `splits` fp32 slabs inside ws */
int rovit_wgrad_splits(int M, int N, int K);
size_t rovit_wgrad_workspace_bytes(int N, int K, int splits);
int rovit_wgrad(const void* dY, int ldy, const void* A, int lda, int M, int N, int K, int splits, int patch_tokens, float* ws,
                rovit_stream_t stream);
/* Up to 4 weight gradients that share M (the four linears of a transformer block) in ONE launch: 96 x 192 output tiles,
 * `splits` M-splits for all of them (24 tiles per split for a DeiT-Tiny block, so 16 splits fill the chip and the fp32
 * partial slabs are 28 MB instead of 75.6 MB per block).  Arrays of n entries; ws[j] sized by
 * rovit_wgrad_workspace_bytes(N[j], K[j], splits) and finished by rovit_wgrad_reduce with the same `splits`. */
int rovit_wgrad_multi(const void* const* dY, const int* ldy, const void* const* A, const int* lda, const int* N, const int* K,
                      float* const* ws, int n, int M, int splits, rovit_stream_t stream);
/* the same with per-problem operand layouts: a_blk[j] / y_blk[j] != 0 = A / dY of problem j is chunk-major [cols / 32][M][32]
 * (rovit_mlp_fused_fwd's act, rovit_mlp_fused_bwd's dpre); NULL = all row-major */
int rovit_wgrad_multi_ex(const void* const* dY, const int* ldy, const void* const* A, const int* lda, const int* N, const int* K,
                         float* const* ws, const int* a_blk, const int* y_blk, int n, int M, int splits, rovit_stream_t stream);
int rovit_wgrad_reduce(const float* ws, int splits, int N, int K, const float* gamma, const float* beta, const float* W,
                       float* dW, float* db, float* dgamma, float* dbeta, float* g_scratch, rovit_stream_t stream);
/* softmax(q k^T * scale) v per (image, head); qkv bf16 (B*T, 3*H*64) = [q|k|v]; out bf16 (B*T, H*64);
 * lse2 (B,H,T) = log2 sum exp(scale q.k) */
int rovit_attention_fwd(const void* qkv, void* out, float* lse2, int batch, int tokens, int heads, int head_dim, float scale,
                        rovit_stream_t stream);
/* softmax(scale q k^T) as fp32 (B,H,T,T) from a saved qkv tensor -- explainability only */
int rovit_attention_probs(const void* qkv, float* probs, int batch, int tokens, int heads, int head_dim, float scale,
                          rovit_stream_t stream);
int rovit_attention_bwd(const void* qkv, const void* out, const float* lse2, const void* dout, void* dqkv, int batch, int tokens,
                        int heads, int head_dim, float scale, rovit_stream_t stream);
/* The same attention when ONLY THE CLASS TOKEN'S output is consumed -- the last block of the backbone, whose other rows nothing reads
 * (timm VisionTransformer.forward_head takes x[:, 0], reached through models/backbone.py:23-25): 197 scores per (image, head) instead
 * of 197 x 197.  Forward writes out[b, 0, :] and lse2[b, h, 0] only.  Backward takes the gradient of that row (dout[b, 0, :]; the other
 * rows of dout are not read) and writes the WHOLE dqkv: dK, dV for every token, dQ for the class token, zeros in every other dQ row. */
int rovit_attention_cls_fwd(const void* qkv, void* out, float* lse2, int batch, int tokens, int heads, int head_dim, float scale,
                            rovit_stream_t stream);
int rovit_attention_cls_bwd(const void* qkv, const void* out, const float* lse2, const void* dout, void* dqkv, int batch, int tokens,
                            int heads, int head_dim, float scale, rovit_stream_t stream);
int rovit_layernorm_fwd(const float* x, void* xhat, float* rstd, int rows, int dim, float eps, rovit_stream_t stream);
int rovit_layernorm_bwd(const void* dxhat, const void* xhat, const float* rstd, float* dX, void* dXb, int rows, int dim,
                        rovit_stream_t stream);
int rovit_im2col(const float* x, void* col, int batch, rovit_stream_t stream);
/* PatchEmbed (timm conv k16 s16, reached through models/backbone.py:12-25) without the im2col buffer: the GEMM and the
 * weight-gradient kernel gather their pixel operand from the fp32 NCHW images and round it to bf16 on the way into LDS.
 *   fwd:   X[b*tokens + 1 + p][:] = patch(b, p) W^T + bias + pos[1 + p]      (W bf16 (192,768), X fp32 (batch*tokens,192))
 *   wgrad: slabs of G[n][k] = sum_(b,p) dY[b*tokens + 1 + p][n] pixel(b,p,k) in ws (rovit_wgrad_workspace_bytes(N,768,splits)),
 *          finished by rovit_wgrad_reduce; dY bf16 (batch*tokens, N). */
int rovit_patch_embed_fwd(const float* images, const void* W, const float* bias, const float* pos, float* X, int batch, int tokens,
                          rovit_stream_t stream);
int rovit_patch_embed_wgrad(const void* dY, int ldy, const float* images, int batch, int tokens, int N, int splits, float* ws,
                            rovit_stream_t stream);
int rovit_cls_rows(const float* cls, const float* pos, float* X, int batch, int tokens, rovit_stream_t stream);
int rovit_cls_norm_fwd(const float* X, const float* gamma, const float* beta, float* feat, float* xhat, float* rstd, int batch,
                       int tokens, float eps, rovit_stream_t stream);
/* backward of the final LayerNorm on the class tokens: writes the CLS rows of dX (fp32) / dXb (bf16); zero_fill != 0 first zeroes both
 * for all tokens (rovit_vit_backward passes 0: only CLS rows are read behind it) */
int rovit_cls_norm_bwd(const float* dfeat, const float* xhat, const float* rstd, const float* gamma, float* dX, void* dXb,
                       float* dgamma, float* dbeta, int batch, int tokens, int zero_fill, rovit_stream_t stream);
/* dpos[t] = sum over images of the token gradient, dcls = dpos[0]; the gradient as fp32 rows (dX) or bf16 rows (dXb): exactly one */
int rovit_pos_grad(const float* dX, const void* dXb, float* dpos, float* dcls, int batch, int tokens, rovit_stream_t stream);
int rovit_prep_weight(const float* W, const float* bias, const float* gamma, const float* beta, void* Wf, void* WfT,
                      float* bias_f, int N, int K, rovit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batch augmentation on the device: the data movement of `cutmix_or_mixup` (call site training/trainer.py:84-96;
 * its module data/transforms.py is absent from the reference checkout, so this follows the published MixUp / CutMix
 * definitions).  mode 0: out[b] = lam x[b] + (1-lam) x[perm[b]];  mode 1: out[b] = x[perm[b]] inside rows [y0,y1) x
 * cols [x0,x1), x[b] elsewhere.  images/out fp32 (B,C,H,W) distinct buffers, perm int64 (B) on the device.
 * ------------------------------------------------------------------------------------------------------------ */
int rovit_mix_images(const float* images, float* out, const long long* perm, int batch, int channels, int height, int width,
                     int mode, float lam, int y0, int y1, int x0, int x1, rovit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer step on flat fp32 buffers: global-norm clipping + AdamW as the reference applies them
 * (training/trainer.py:123-128,137-141 clip_grad_norm_(1.0); training/optimizer.py:7-32 AdamW).
 * rovit_sq_norm_accum: *out_sq += sum g^2 (caller zeroes out_sq; several buffers may accumulate into one norm).
 *   scratch: NULL, or >= 520 floats zeroed ONCE by the caller and then owned by this function (block partials are then
 *   combined in a fixed order: bit-reproducible; with NULL they are combined with float atomics).
 * rovit_adamw_flat: torch.optim.AdamW semantics; grad_scale = device scalar multiplied into g (clip coefficient)
 * or NULL; t = 1-based step count for the bias correction.
 * ------------------------------------------------------------------------------------------------------------ */
int rovit_sq_norm_accum(const float* g, size_t n, float* out_sq, float* scratch, rovit_stream_t stream);
/* clip_grad_norm_ coefficient on the device: *norm_out = sqrt(*sq) (optional), *coef = min(1, max_norm / (norm + 1e-6)) */
int rovit_clip_coef(const float* sq, float max_norm, float* coef, float* norm_out, rovit_stream_t stream);
int rovit_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, const float* grad_scale, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int t, rovit_stream_t stream);
/* The same two steps as ONE launch each (round 4).  rovit_sq_norm_clip: squared norm over up to four buffers (HOST arrays bufs /
 * counts; 16-byte aligned, counts multiples of 4), block partials added in block order by the block that finishes last, then the
 * coefficient; scratch >= 264 floats, zeroed ONCE by the caller (the kernel re-arms its ticket).
 * rovit_adamw_flat_multi: rovit_adamw_flat over up to four segments with their own lr and step count t (HOST arrays). */
int rovit_sq_norm_clip(const float* const* bufs, const size_t* counts, int n_bufs, float max_norm, float* coef, float* norm_out,
                       float* scratch, size_t scratch_floats, rovit_stream_t stream);
int rovit_adamw_flat_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const size_t* n, const float* lr,
                           const int* t, int n_segs, const float* grad_scale, float beta1, float beta2, float eps, float weight_decay,
                           rovit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Joint multi-task loss, forward + gradient in one launch: JointLoss.forward (training/losses.py:139-181) with
 * FocalLoss (:15-38), OrdinalBCELoss (:48-72), UncertaintyLoss (:80-101), KANRegressionLoss (:109-114).
 * Class targets are int64 (torch.long); severity targets fp32, or int64 labels with severity_is_int64 != 0 (the reference casts
 * them with .float(), :89-90, :110-111 -- done inside the kernel; ordinal targets are (severity > k), :55-56).  A class label outside [0, num_classes) makes every loss NaN
 * instead of reading out of bounds.  NULL head pointers = head inactive at this curriculum stage.
 * d_* = d(total)/d(head output) for an upstream gradient of 1; losses_out = [cls, ord, unc, kan, total].
 * rovit_scale_buffers multiplies up to 5 buffers by a device scalar (chain rule with the upstream gradient).
 * ------------------------------------------------------------------------------------------------------------ */
int rovit_joint_loss(const float* cls_logits, const float* ordinal_logits, const float* mu, const float* log_var,
                     const float* kan_severity, const long long* class_targets, const void* severity_targets, int severity_is_int64,
                     const float* focal_alpha, float* d_cls, float* d_ord, float* d_mu, float* d_lv, float* d_kan,
                     float* losses_out, int batch, int num_classes, float lambda_ord, float mu_unc, float nu_kan, float focal_gamma,
                     rovit_stream_t stream);
int rovit_scale_buffers(float* const* bufs, const int* counts, int n, const float* scale, rovit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ROVIT_HIP_H */
