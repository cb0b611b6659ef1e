"""Explainability taps from the fused backbone."""
from typing import List

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
