"""Optimizer for the HIP path: global-norm gradient clipping + AdamW with the reference's parameter grouping.

Reference: training/optimizer.py:7-32 (AdamW; parameters whose name contains 'backbone' at lr/10, the rest at lr;
weight_decay 1e-4) and training/trainer.py:123-128,137-141 (clip_grad_norm_(parameters, 1.0) before the step).

The backbone's 5.5 M parameters are re-homed into ONE flat fp32 buffer (each nn.Parameter keeps its identity and
becomes a view), next to the flat gradient buffer the HIP backward already writes, so their whole update is two
launches (rovit_sq_norm_accum + rovit_adamw_flat).  The ~0.18 M head/KAN parameters keep torch.optim.AdamW.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import native
from .native import call, ptr, stream_ptr


class RoViTAdamW:
    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: Optional[float] = 1.0):
        self.model = model
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.max_grad_norm = max_grad_norm
        self.vit = model.backbone.model
        self.engine = self.vit.engine
        self.bb_params = self.vit.ordered_parameters()
        dev = self.bb_params[0].device
        if dev.type != 'cuda':
            raise native.RovitHipError('RoViTAdamW needs the model on a CUDA/HIP device (call model.to(device) first)')
        total = sum(p.numel() for p in self.bb_params)
        self.p_flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.bb_params:                        # re-home every backbone parameter into the flat buffer
            n = p.numel()
            view = self.p_flat[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            off += n
        self.m_flat = torch.zeros_like(self.p_flat)
        self.v_flat = torch.zeros_like(self.p_flat)
        self.t = 0
        self.other_params = [p for n, p in model.named_parameters() if not n.startswith('backbone.')]
        self.other = torch.optim.AdamW(self.other_params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        self._sq = torch.zeros((), dtype=torch.float32, device=dev)
        self.last_grad_norm: Optional[torch.Tensor] = None

    def zero_grad(self, set_to_none: bool = True):
        for p in self.bb_params + self.other_params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _backbone_active(self) -> bool:
        return self.bb_params[0].requires_grad and self.bb_params[0].grad is not None

    @torch.no_grad()
    def step(self):
        eng = self.engine
        bb = self._backbone_active()
        if bb and (eng.grad_views is None or self.bb_params[0].grad.data_ptr() != eng.grad_views[0].data_ptr()):
            raise native.RovitHipError('backbone gradients are not the engine-owned flat buffer')
        others = [p for p in self.other_params if p.grad is not None]
        scale = None
        if self.max_grad_norm is not None:
            self._sq.zero_()
            if bb:
                call('rovit_sq_norm_accum', ptr(eng.grad_flat), eng.grad_flat.numel(), ptr(self._sq), stream_ptr())
            if others:
                norms = torch._foreach_norm([p.grad for p in others])
                self._sq.add_(torch.stack(norms).square_().sum())
            total_norm = self._sq.sqrt()
            self.last_grad_norm = total_norm
            scale = (self.max_grad_norm / (total_norm + 1e-6)).clamp_(max=1.0)    # clip_grad_norm_ semantics
            if others:
                torch._foreach_mul_([p.grad for p in others], scale)
        self.t += 1
        if bb:
            call('rovit_adamw_flat', ptr(self.p_flat), ptr(eng.grad_flat), ptr(self.m_flat), ptr(self.v_flat),
                 self.p_flat.numel(), ptr(scale) if scale is not None else None, self.lr / 10.0, self.betas[0], self.betas[1],
                 self.eps, self.wd, self.t, stream_ptr())
            eng._prep_key = None            # parameters changed behind torch's version counters: re-prepare weights
        if others:
            self.other.step()
