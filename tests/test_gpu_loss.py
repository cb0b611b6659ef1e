"""GPU parity of the fused JointLoss kernel against the vectors generated from the reference's training/losses.py
(tests/golden/joint_loss.npz) and against the tensor-op restatement on random batches."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def dev():
    return torch.device('cuda:0')


def test_fused_joint_loss_vs_reference_golden(golden_dir):
    from rovit_hip.losses import JointLoss
    g = np.load(os.path.join(golden_dir, 'joint_loss.npz'))
    y, alpha = T(g['y']).to(dev()), T(g['alpha']).to(dev())
    for stage in (1, 2, 3, 4):
        outd = {k[3:]: T(g[k]).to(dev()).requires_grad_(True) for k in g.files if k.startswith('in.')}
        l = JointLoss(1.0, 0.5, 0.5, 2.0, alpha)(outd, y, y, stage)
        for k in ('cls_loss', 'ord_loss', 'unc_loss', 'kan_loss', 'total_loss'):
            assert abs(float(l[k].detach()) - float(g[f's{stage}.{k}'])) < 2e-5, (stage, k)
        (3.0 * l['total_loss']).backward()                      # non-trivial upstream gradient
        for k, v in outd.items():
            got = v.grad.cpu() if v.grad is not None else torch.zeros(v.shape)
            assert float((got - 3.0 * T(g[f's{stage}.grad.{k}'])).abs().max()) < 3e-6, (stage, k)


@pytest.mark.parametrize('B', [1, 7, 256, 1000])
def test_fused_joint_loss_vs_tensor_ops(B):
    from rovit_hip.losses import JointLoss
    torch.manual_seed(B)
    mk = lambda *s: (torch.randn(*s, device=dev()) * 2).requires_grad_(True)
    a = {'cls_logits': mk(B, 4), 'ordinal_logits': mk(B, 3), 'mu': mk(B, 1), 'log_var': mk(B, 1),
         'kan_severity': (3 * torch.rand(B, 1, device=dev())).requires_grad_(True)}
    b = {k: v.detach().clone().requires_grad_(True) for k, v in a.items()}
    yc = torch.randint(0, 4, (B,), device=dev())
    ys = torch.randint(0, 4, (B,), device=dev())
    lf = JointLoss(1.0, 0.5, 0.5, 2.0, None)
    la = lf(a, yc, ys, 4)
    lb = lf._forward_tensor_ops(b['cls_logits'], b['ordinal_logits'], b['mu'], b['log_var'], b['kan_severity'], yc, ys)
    assert abs(float(la['total_loss'].detach()) - float(lb['total_loss'].detach())) < 1e-4 * max(1.0, abs(float(lb['total_loss'].detach())))
    la['total_loss'].backward()
    lb['total_loss'].backward()
    for k in a:
        assert float((a[k].grad - b[k].grad).abs().max()) < 1e-5 * max(1.0, float(b[k].grad.abs().max())), k
