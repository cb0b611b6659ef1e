"""Joint multi-task loss that seeds the backward of the hot path.

Mirrors /root/reference/training/losses.py JointLoss (:117-181; FocalLoss :7-38, OrdinalBCELoss :41-72,
UncertaintyLoss :75-101, KANRegressionLoss :104-114): same constructor, same call signature, same returned dict.
On CUDA/HIP tensors the whole loss and its gradient are ONE HIP launch (rovit_joint_loss; backward = one scale by
the upstream gradient).  CPU tensors use the plain-tensor-op restatement below (same formulas; used by the CPU unit
tests of the host logic -- the head outputs of the HIP model are always device tensors).
"""
from typing import Dict, Optional

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .native import call, ptr, stream_ptr


class JointLossFn(torch.autograd.Function):
    """inputs: cls, ord|None, mu|None, lv|None, kan|None, class_t, sev_t, alpha|None, (lambda, mu, nu, gamma)
    outputs: total (0-dim, differentiable), components (4,) detached [cls, ord, unc, kan]."""

    @staticmethod
    def forward(ctx, cls, ordl, mu, lv, kan, cls_t, sev_t, alpha, weights):
        f = lambda t: None if t is None else t.detach().float().contiguous()
        cls, ordl, mu, lv, kan = f(cls), f(ordl), f(mu), f(lv), f(kan)
        cls_t = cls_t.long().contiguous()
        # severity is used as float (reference losses.py:89-90,110-111): int64 labels are converted inside the kernel, anything else here
        sev_t = sev_t.detach().reshape(-1)
        sev_i64 = sev_t.dtype == torch.int64
        sev_t = (sev_t if sev_i64 else sev_t.float()).contiguous()
        alpha = f(alpha.to(cls.device)) if alpha is not None else None
        B, C = cls.shape
        out = torch.empty(5, device=cls.device, dtype=torch.float32)
        grads = [torch.empty_like(t) if t is not None else None for t in (cls, ordl, mu, lv, kan)]
        lam, muw, nu, gamma = weights
        call('rovit_joint_loss', ptr(cls), ptr(ordl), ptr(mu), ptr(lv), ptr(kan), ptr(cls_t), ptr(sev_t), int(sev_i64), ptr(alpha),
             ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(grads[3]), ptr(grads[4]), ptr(out), B, C, lam, muw, nu, gamma,
             stream_ptr())
        ctx.grads = grads
        ctx.set_materialize_grads(False)      # no zeros(4) launch for the (non-differentiable) components' gradient
        comps = out[:4]
        ctx.mark_non_differentiable(comps)
        return out[4], comps

    @staticmethod
    def backward(ctx, g_total, _g_comps):
        if ctx.grads is None:
            raise native.RovitHipError('JointLoss: backward called twice on the same graph (the fused kernel scales its '
                                       'gradient buffers in place); call the loss again instead of retain_graph=True')
        if g_total is None:
            return (None,) * 9
        grads, ctx.grads = ctx.grads, None
        live = [g for g in grads if g is not None]
        arr = (ctypes.c_void_p * len(live))(*[g.data_ptr() for g in live])
        cnt = (ctypes.c_int * len(live))(*[g.numel() for g in live])
        scale = g_total.detach().float().contiguous()
        call('rovit_scale_buffers', arr, cnt, len(live), ptr(scale), stream_ptr())
        return (grads[0], grads[1], grads[2], grads[3], grads[4], None, None, None, None)


class JointLoss(nn.Module):
    def __init__(self, lambda_ord: float = 1.0, mu_unc: float = 0.5, nu_kan: float = 0.5, focal_gamma: float = 2.0,
                 focal_alpha: Optional[torch.Tensor] = None, num_classes: int = 4):
        super().__init__()
        self.lambda_ord, self.mu_unc, self.nu_kan = lambda_ord, mu_unc, nu_kan
        self.focal_gamma, self.num_classes = focal_gamma, num_classes
        self.focal_alpha = focal_alpha

    def forward(self, outputs: Dict[str, torch.Tensor], class_targets: torch.Tensor, severity_targets: torch.Tensor,
                stage: int = 4) -> Dict[str, torch.Tensor]:
        logits = outputs['cls_logits']
        ordl = outputs['ordinal_logits'] if stage >= 2 else None
        mu = outputs['mu'] if stage >= 3 else None
        lv = outputs['log_var'] if stage >= 3 else None
        if mu is None or lv is None:
            mu = lv = None
        kan = outputs['kan_severity'] if stage >= 4 else None
        if logits.is_cuda:
            total, comps = JointLossFn.apply(logits, ordl, mu, lv, kan, class_targets, severity_targets, self.focal_alpha,
                                             (self.lambda_ord, self.mu_unc, self.nu_kan, self.focal_gamma))
            return {'cls_loss': comps[0], 'ord_loss': comps[1], 'unc_loss': comps[2], 'kan_loss': comps[3], 'total_loss': total}
        return self._forward_tensor_ops(logits, ordl, mu, lv, kan, class_targets, severity_targets)

    def _forward_tensor_ops(self, logits, ordl, mu, lv, kan, class_targets, severity_targets):
        logp = F.log_softmax(logits, dim=1)
        lp_t = logp.gather(1, class_targets.unsqueeze(1)).squeeze(1)
        focal = (1.0 - lp_t.exp()) ** self.focal_gamma * (-lp_t)
        if self.focal_alpha is not None:
            focal = self.focal_alpha.to(logits.device)[class_targets] * focal
        losses = {'cls_loss': focal.mean()}
        total = losses['cls_loss']
        zero = torch.zeros((), device=logits.device)
        sev = severity_targets.float().unsqueeze(1) if severity_targets.dim() == 1 else severity_targets.float()
        losses['ord_loss'] = losses['unc_loss'] = losses['kan_loss'] = zero
        if ordl is not None:
            thr = torch.arange(ordl.shape[1], device=ordl.device)
            bt = (sev.reshape(-1, 1) > thr).float()
            losses['ord_loss'] = F.binary_cross_entropy_with_logits(ordl, bt, reduction='none').mean(dim=1).mean()
            total = total + self.lambda_ord * losses['ord_loss']
        if mu is not None:
            losses['unc_loss'] = (0.5 * ((sev - mu) ** 2 * torch.exp(-lv) + lv)).mean()
            total = total + self.mu_unc * losses['unc_loss']
        if kan is not None:
            losses['kan_loss'] = F.mse_loss(kan, sev)
            total = total + self.nu_kan * losses['kan_loss']
        losses['total_loss'] = total
        return losses
