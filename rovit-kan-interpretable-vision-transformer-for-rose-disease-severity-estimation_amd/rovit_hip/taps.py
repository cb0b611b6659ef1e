"""Explainability taps from the fused backbone."""
import ctypes
from typing import List, Sequence

import torch

from . import native
from .native import call, ptr, ptr_array, stream_ptr


def attention_outputs(model, x: torch.Tensor) -> List[torch.Tensor]:
    """What the reference's DeiTTinyBackbone.get_attention_maps returns (models/backbone.py:37-62): one tensor per
    block holding the OUTPUT of that block's attention module, shape (B, 197, 192) (on current timm the forward hook
    on ``block.attn`` sees the module output, not the attention probabilities -- SURVEY.md 8(a) row a4).
    ``model`` is the DeiTTiny parameter container (``backbone.model``)."""
    x = x.float().contiguous()
    params = model.ordered_parameters()
    eng = model.engine
    eng.prepare(params)
    B = x.shape[0]
    ws = eng.take_ws(B, False, x.device)
    feats = torch.empty(B, 192, device=x.device, dtype=torch.float32)
    taps = [torch.empty(B * 197, 192, device=x.device, dtype=torch.bfloat16) for _ in range(eng.depth)]
    with torch.no_grad():
        call('rovit_vit_forward_taps', ptr(x), ptr_array(params), ptr(eng.prep), ptr(ws), ptr(feats), ptr_array(taps), None, B,
             eng.depth, stream_ptr())
    eng.give_ws(B, False, ws)
    return [t.float().view(B, 197, 192) for t in taps]


def attention_probabilities(model, x: torch.Tensor) -> List[torch.Tensor]:
    """The softmax attention probabilities of every block, (B, 3, 197, 197) fp32 -- what the reference's rollout code
    (explainability/attention_maps.py:18-105) means to collect from its hooks (current timm no longer returns them,
    SURVEY.md 8(a) row a4 / 8(f) row f-4).  ``model`` is the DeiTTiny parameter container (``backbone.model``)."""
    x = x.float().contiguous()
    params = model.ordered_parameters()
    eng = model.engine
    eng.prepare(params)
    B = x.shape[0]
    ws = eng.take_ws(B, False, x.device)
    feats = torch.empty(B, 192, device=x.device, dtype=torch.float32)
    probs = [torch.empty(B, 3, 197, 197, device=x.device, dtype=torch.float32) for _ in range(eng.depth)]
    with torch.no_grad():
        call('rovit_vit_forward_taps', ptr(x), ptr_array(params), ptr(eng.prep), ptr(ws), ptr(feats), None, ptr_array(probs), B,
             eng.depth, stream_ptr())
    eng.give_ws(B, False, ws)
    return probs


# ---- views into the training workspace (include/rovit_hip.h: rovit_vit_workspace_field) ------------------------
WS_XHAT1, WS_RSTD1, WS_QKV, WS_ATTN_O, WS_XHAT2, WS_RSTD2, WS_ACT, WS_DQKV = range(8)
_FIELD_DTYPE = {WS_XHAT1: (torch.bfloat16, 192), WS_QKV: (torch.bfloat16, 576), WS_ATTN_O: (torch.bfloat16, 192),
                WS_XHAT2: (torch.bfloat16, 192), WS_ACT: (torch.bfloat16, 768), WS_DQKV: (torch.bfloat16, 576),
                WS_RSTD1: (torch.float32, 1), WS_RSTD2: (torch.float32, 1)}


def workspace_view(ws: torch.Tensor, batch: int, depth: int, field: int, block: int, mlp_path: int = native.MLP_AUTO) -> torch.Tensor:
    """(M, width) view of one saved buffer of a training workspace -- zero-copy, except WS_ACT when the one-launch MLP half wrote it
    (mlp_path of the forward, see include/rovit_hip.h): that buffer is CHUNK-MAJOR [24][M][32] and is de-interleaved into a row-major
    copy here (round 3 returned the raw bytes under a row-major shape)."""
    off, nbytes = ctypes.c_size_t(), ctypes.c_size_t()
    call('rovit_vit_workspace_field', batch, depth, field, block, ctypes.byref(off), ctypes.byref(nbytes))
    dt, width = _FIELD_DTYPE[field]
    M = batch * 197
    flat = ws[off.value:off.value + nbytes.value].view(dt)
    if field == WS_ACT and (mlp_path == native.MLP_ONE_LAUNCH or (mlp_path == native.MLP_AUTO and M >= native.MLP_FUSED_MIN_ROWS)) \
            and block != depth - 1:            # (the last block's MLP half runs on the CLS rows with the two-launch kernels: row-major)
        return flat.view(24, M, 32).permute(1, 0, 2).reshape(M, width)
    return flat.view(M, width)


def norm1_output(model, block: int) -> torch.Tensor:
    """Output of ``blocks[block].norm1`` (B,197,192) of the most recent grad-mode forward: the kernels keep the
    normalised rows xhat (bf16) for the backward; the affine is applied here (what a forward hook on norm1 sees,
    reference explainability/gradcam.py:18-20,40)."""
    eng = model.engine
    if eng.last_ws is None:
        raise native.RovitHipError('norm1_output: no training-mode forward is pending (run the model with grad enabled first)')
    ws, B = eng.last_ws[:2]
    blk = model.blocks[block]
    xhat = workspace_view(ws, B, eng.depth, WS_XHAT1, block)
    return (xhat.float() * blk.norm1.weight.detach() + blk.norm1.bias.detach()).view(B, 197, 192)


def norm1_output_grad(params: Sequence[torch.Tensor], ws: torch.Tensor, batch: int, depth: int, block: int) -> torch.Tensor:
    """dL/d(output of blocks[block].norm1), (B,197,192): the qkv dgrad against the UNFOLDED weight, computed by the
    library's own GEMM from the dqkv buffer the fused backward has just written (valid right after
    rovit_vit_backward(first_block=.., last_block=block); what a full-backward hook on norm1 sees, gradcam.py:22-26)."""
    qkv_w = params[6 + 12 * block + 2]                       # ordered_parameters(): block i = [n1w, n1b, qkvw, ...]
    wt = qkv_w.detach().t().contiguous().to(torch.bfloat16)  # (192, 576): row n holds d(qkv[:])/d(y[n])
    dqkv = workspace_view(ws, batch, depth, WS_DQKV, block)
    out = torch.empty(batch * 197, 192, device=ws.device, dtype=torch.bfloat16)
    call('rovit_gemm_nt', ptr(dqkv), 576, ptr(wt), 576, batch * 197, 192, 576, None, 0, ptr(out), 192, None, None, 0, None, 0,
         None, 0, stream_ptr())
    return out.float().view(batch, 197, 192)
