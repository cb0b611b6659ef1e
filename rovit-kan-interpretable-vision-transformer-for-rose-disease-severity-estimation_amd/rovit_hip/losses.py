"""Joint multi-task loss that seeds the backward of the hot path.

Restates /root/reference/training/losses.py (FocalLoss :15-38, OrdinalBCELoss :48-72, UncertaintyLoss :80-101,
KANRegressionLoss :109-114, JointLoss.forward :139-181).  It is O(B x 4) work on the head outputs; SURVEY.md
section 8 lists it as "next" row f-1.  This module is the host-side mirror of that interface (same constructor,
same returned dict) written with device-agnostic tensor ops; the fused HIP version replaces it behind the same
class when row f-1 is built.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


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
        if stage >= 2 and outputs['ordinal_logits'] is not None:
            ol = outputs['ordinal_logits']
            thr = torch.arange(ol.shape[1], device=ol.device)
            bt = (severity_targets.unsqueeze(1) > thr).float()
            losses['ord_loss'] = F.binary_cross_entropy_with_logits(ol, bt, reduction='none').mean(dim=1).mean()
            total = total + self.lambda_ord * losses['ord_loss']
        if stage >= 3 and outputs['mu'] is not None and outputs['log_var'] is not None:
            mu, lv = outputs['mu'], outputs['log_var']
            losses['unc_loss'] = (0.5 * ((sev - mu) ** 2 * torch.exp(-lv) + lv)).mean()
            total = total + self.mu_unc * losses['unc_loss']
        if stage >= 4 and outputs['kan_severity'] is not None:
            losses['kan_loss'] = F.mse_loss(outputs['kan_severity'], sev)
            total = total + self.nu_kan * losses['kan_loss']
        losses['total_loss'] = total
        return losses
