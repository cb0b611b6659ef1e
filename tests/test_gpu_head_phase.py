"""GPU parity tests of the fused head phase (csrc/head_phase.hip: rovit_head_phase_fwd / _bwd, round 4): the three heads and the
KAN stack as ONE forward launch and a two-launch backward, against
  * the CPU oracle (oracle/ref_cpu.py heads_forward / kan_module_forward, which are pinned to the imported reference by
    tests/golden/, and torch autograd of the same for the gradients), and
  * the per-module HIP path (HeadsFn + KANStackFn) the fused path replaces inside RoViTKAN.forward.
Everything here is fp32: tolerance 2e-5 on outputs of O(1), 1e-4 relative on gradients (summation order differs).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu  # noqa: E402  (checker only)


def dev():
    return torch.device('cuda:0')


def _state(embed, hid, ncls, kan_layers, num_knots, seed):
    g = torch.Generator().manual_seed(seed)
    sd = ref_cpu.init_heads_state(embed, hid, ncls, g)
    sd.update(ref_cpu.init_kan_state(list(kan_layers), num_knots, 3, g, prefix='kan_module.'))
    return sd, g


HEAD_KEYS = ['classification_head.fc1.weight', 'classification_head.fc1.bias', 'classification_head.fc2.weight',
             'classification_head.fc2.bias', 'ordinal_head.fc1.weight', 'ordinal_head.fc1.bias', 'ordinal_head.fc2.weight',
             'ordinal_head.fc2.bias', 'uncertainty_head.fc1.weight', 'uncertainty_head.fc1.bias', 'uncertainty_head.fc_mu.weight',
             'uncertainty_head.fc_mu.bias', 'uncertainty_head.fc_logvar.weight', 'uncertainty_head.fc_logvar.bias']


def _kan_keys(n):
    out = []
    for l in range(n):
        out += [f'kan_module.kan_layers.{l}.spline_weights', f'kan_module.kan_layers.{l}.linear.weight', f'kan_module.kan_layers.{l}.linear.bias']
    return out


def _oracle(features, sd, stage, masks=None):
    out = ref_cpu.heads_forward(features, sd, stage, masks)
    out['kan'] = ref_cpu.kan_module_forward(features, sd, prefix='kan_module.') if stage >= 4 else None
    return out


def _fused(features, sd, stage, kan_layers, masks=None, drop=None, requires_grad=True):
    from rovit_hip.functions import ACT_RELU, ACT_SIGMOID3, HeadPhaseFn
    nl = len(kan_layers) - 1
    keys = HEAD_KEYS + (_kan_keys(nl) if stage >= 4 else [])
    params = [sd[k].to(dev()).clone().requires_grad_(requires_grad) for k in keys]
    cfg = {'stage': stage, 'masks': masks, 'drop_p': 0.0, 'seed': 0, 'offset': 0,
           'kan_dims': list(kan_layers) if stage >= 4 else [],
           'kan_knots': [sd[f'kan_module.kan_layers.{l}.knots'].to(dev()) for l in range(nl)] if stage >= 4 else [],
           'kan_acts': [ACT_SIGMOID3 if i == nl - 1 else ACT_RELU for i in range(nl)], 'grad_views': None}
    if drop is not None:
        cfg.update(drop_p=drop[0], seed=drop[1], offset=drop[2])
    f = features.to(dev()).clone().requires_grad_(requires_grad)
    outs = HeadPhaseFn.apply(f, cfg, *params)
    return f, params, keys, outs


def _weighted(outs, ws):
    return sum((o * w).sum() for o, w in zip(outs, ws) if o is not None and o.numel())


@pytest.mark.parametrize('stage', [1, 2, 3, 4])
@pytest.mark.parametrize('cfg', [
    dict(B=37, embed=192, hid=128, ncls=4, kan=(192, 64, 16, 1), knots=5),       # the reference's head phase (num_basis 7: dense-row kernels)
    dict(B=5, embed=192, hid=128, ncls=4, kan=(192, 64, 16, 1), knots=32),       # BASELINE configs[4] (KAN-heavy): gathered-weight kernels
    dict(B=9, embed=192, hid=128, ncls=4, kan=(192, 64, 16, 1), knots=6),        # num_basis 8: aligned dense rows
    dict(B=3, embed=384, hid=64, ncls=5, kan=(384, 48, 8, 4, 1), knots=5),       # four layers, widths that do not divide the workgroup
    dict(B=4, embed=96, hid=32, ncls=3, kan=(96, 1), knots=9),                   # one layer
], ids=['reference', 'g32', 'nb8', 'four_layers', 'one_layer'])
def test_head_phase_forward_and_backward_vs_oracle(cfg, stage):
    sd, g = _state(cfg['embed'], cfg['hid'], cfg['ncls'], cfg['kan'], cfg['knots'], seed=1000 + cfg['B'] + stage)
    B = cfg['B']
    x = torch.randn(B, cfg['embed'], generator=g)
    x[0, :8] = torch.tensor([-30., 30., -4., 4., 0., 1e-8, 2.5, -2.5])             # saturated / edge inputs of the basis
    masks_cpu = {k: (torch.rand(B, cfg['hid'], generator=g) < 0.7).float() / 0.7 for k in ('cls', 'ord', 'unc')}
    ws = [torch.randn(B, n, generator=g) for n in (cfg['ncls'], cfg['ncls'] - 1, 1, 1, cfg['kan'][-1])]
    # oracle
    xr = x.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and 'knots' not in k) for k, v in sd.items()}
    o = _oracle(xr, sdr, stage, masks_cpu)
    ref_outs = [o['cls_logits'], o['ordinal_logits'], o['mu'], o['log_var'], o['kan']]
    _weighted(ref_outs, ws).backward()
    # fused HIP path
    masks = [masks_cpu['cls'].to(dev()), masks_cpu['ord'].to(dev()), masks_cpu['unc'].to(dev())]
    f, params, keys, outs = _fused(x, sd, stage, cfg['kan'], masks)
    for got, ref in zip(outs, ref_outs):
        if ref is None:
            assert got.numel() == 0
        else:
            assert got.shape == ref.shape
            assert float((got.detach().cpu() - ref.detach()).abs().max()) < 2e-5
    _weighted(outs, [w.to(dev()) for w in ws]).backward()

    # gradients: 1e-4 of the tensor's largest entry; 3e-4 at num_knots 32, where the basis derivative carries 1 / h = 18.5 and the sums
    # over 192 x 64 terms of mixed sign lose four digits in fp32 either way (the fp32 CPU oracle is as far from an fp64 evaluation)
    rtol = 3e-4 if cfg['knots'] >= 32 else 1e-4

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-6
        assert float((a.cpu() - b).abs().max()) <= rtol * scale + 1e-7, what
    close(f.grad, xr.grad, 'd_features')
    nheads = 3 if stage >= 3 else (2 if stage >= 2 else 1)
    for k, p in zip(keys, params):
        head = {'classification_head': 0, 'ordinal_head': 1, 'uncertainty_head': 2}.get(k.split('.')[0])
        if head is not None and head >= nheads:
            assert p.grad is None, k                      # gated head: no gradient, like the reference's unused modules
            continue
        close(p.grad, sdr[k].grad, k)


def test_head_phase_equals_the_per_module_path_inside_the_model():
    """RoViTKAN.forward through the fused launch and through HeadsFn + KANStackFn: same outputs, same gradients (eval mode)."""
    from models.rovit_kan import RoViTKAN
    torch.manual_seed(3)
    m = RoViTKAN(pretrained=False).to(dev()).eval()
    feats = torch.randn(33, 192, device=dev())
    res = {}
    for name, limit in (('fused', 1024), ('modules', 0)):
        m.head_phase_max_batch = limit
        for p in m.parameters():
            p.grad = None
        f = feats.clone().requires_grad_(True)
        assert m._head_phase_fusable(f) == (name == 'fused')
        if name == 'fused':
            out = m._forward_head_phase(f, 4)
        else:
            from rovit_hip.functions import HeadsFn
            c, o, mu, lv = HeadsFn.apply(f, 4, None, *m._head_params())
            out = {'cls_logits': c, 'ordinal_logits': o, 'mu': mu, 'log_var': lv, 'kan_severity': m.kan_module(f)}
        loss = sum(out[k].square().sum() * w for k, w in (('cls_logits', 1.0), ('ordinal_logits', 0.5), ('mu', 0.3), ('log_var', 0.2),
                                                             ('kan_severity', 2.0)))
        loss.backward()
        res[name] = ({k: out[k].detach().clone() for k in ('cls_logits', 'ordinal_logits', 'mu', 'log_var', 'kan_severity')}, f.grad.clone(),
                     {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    m.head_phase_max_batch = 1024
    for k in res['fused'][0]:
        assert float((res['fused'][0][k] - res['modules'][0][k]).abs().max()) < 2e-5, k
    scale = float(res['modules'][1].abs().max())
    assert float((res['fused'][1] - res['modules'][1]).abs().max()) < 1e-4 * scale
    assert set(res['fused'][2]) == set(res['modules'][2]) and len(res['fused'][2]) == 14 + 9
    for n, gm in res['modules'][2].items():
        assert float((res['fused'][2][n] - gm).abs().max()) <= 1e-4 * float(gm.abs().max()) + 1e-7, n


def test_dropout_drawn_in_the_kernel_statistics_determinism_and_backward():
    """Training mode without mask tensors: the keep decisions come from Philox inside the forward kernel.  Kept share ~ 1 - p,
    kept units carry relu(pre) / (1 - p), the same (seed, offset) reproduces the draw, another offset does not, and the backward
    (which sees no mask, only the stored hidden values) equals the explicit-mask path run with the mask recovered from the forward."""
    from rovit_hip.functions import HeadPhaseFn  # noqa: F401
    B, E, hid, p = 256, 192, 128, 0.3
    kan = (192, 64, 16, 1)
    sd, g = _state(E, hid, 4, kan, 5, seed=77)
    x = torch.randn(B, E, generator=g)
    f0, _, _, outs0 = _fused(x, sd, 4, kan, None, None, requires_grad=False)               # no dropout
    f1, params1, keys, outs1 = _fused(x, sd, 4, kan, None, (p, 1234, 40))
    f2, _, _, outs2 = _fused(x, sd, 4, kan, None, (p, 1234, 40), requires_grad=False)
    f3, _, _, outs3 = _fused(x, sd, 4, kan, None, (p, 1234, 44), requires_grad=False)
    assert torch.equal(outs1[0], outs2[0]) and torch.equal(outs1[2], outs2[2])
    assert not torch.equal(outs1[0], outs3[0])
    assert torch.equal(outs0[4], outs1[4])                                                   # the KAN branch has no dropout
    # recover the masks: hidden activations are not returned, so use a second explicit-mask run and compare END results instead;
    # first the statistics, from the classification logits' sensitivity: run the heads' first layer on the host
    h_ref = torch.relu(x @ sd['classification_head.fc1.weight'].t() + sd['classification_head.fc1.bias'])      # (B, hid)
    # mask recovery through a probe: a model whose fc2 is the identity on unit k is expensive; instead re-derive the mask the kernel
    # must have used from its documented counter layout and check the outputs against the oracle with THAT mask
    masks = _philox_masks(B, hid, p, 1234, 40)
    share = float((masks[0] > 0).float().mean())
    assert abs(share - (1 - p)) < 4 * math.sqrt(p * (1 - p) / (B * hid))
    o = _oracle(x, sd, 4, {'cls': masks[0], 'ord': masks[1], 'unc': masks[2]})
    for got, ref in zip(outs1[:4], (o['cls_logits'], o['ordinal_logits'], o['mu'], o['log_var'])):
        assert float((got.detach().cpu() - ref).abs().max()) < 2e-5
    assert float((h_ref > 0).float().mean()) > 0.2                                           # the probe is not vacuous
    # backward: in-kernel draw vs the explicit-mask path with the same mask
    ws = [torch.randn(B, n, generator=g).to(dev()) for n in (4, 3, 1, 1, 1)]
    _weighted(outs1, ws).backward()
    fe, params_e, _, outs_e = _fused(x, sd, 4, kan, [m.to(dev()) for m in masks], None)
    _weighted(outs_e, ws).backward()
    assert float((f1.grad - fe.grad).abs().max()) <= 1e-5 * float(fe.grad.abs().max())
    for k, a, b in zip(keys, params1, params_e):
        assert float((a.grad - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max()) + 1e-8, k


def _philox_masks(B, hid, p, seed, offset):
    """The kernel's draw restated on the host (oracle/philox.py, itself pinned to the published Philox4x32-10 known-answer vectors by
    tests/test_oracle.py)."""
    from oracle.philox import head_phase_masks
    return [torch.from_numpy(m) for m in head_phase_masks(B, hid, p, seed, offset)]


def test_training_step_uses_the_fused_phase_and_writes_gradients_into_the_optimizers_flat_buffer():
    """RoViTKAN in train mode + RoViTAdamW: the head / KAN gradients ARE views of the optimizer's flat buffer after backward (no pack
    copy), accumulate correctly over two backward passes, and a model with a forward hook on a head takes the per-module path."""
    from models.rovit_kan import RoViTKAN
    from rovit_hip.losses import JointLoss
    from rovit_hip.optim import RoViTAdamW
    torch.manual_seed(0)
    m = RoViTKAN(pretrained=False, dropout=0.0).to(dev()).train()
    opt = RoViTAdamW(m, lr=1e-3)
    loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0)
    x = torch.randn(4, 3, 224, 224, device=dev())
    y = torch.randint(0, 4, (4,), device=dev())
    views = m._head_grad_views
    loss_fn(m(x), y, y, 4)['total_loss'].backward()
    others = [p for n, p in m.named_parameters() if not n.startswith('backbone.')]
    assert len(others) == 23 and all(p.grad is not None and p.grad.data_ptr() == views[p.data_ptr()].data_ptr() for p in others)
    g1 = [p.grad.clone() for p in others]
    loss_fn(m(x), y, y, 4)['total_loss'].backward()                     # accumulation: not the direct path, grads double
    for p, g in zip(others, g1):
        assert float((p.grad - 2 * g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-9
    before = [p.detach().clone() for p in others]
    opt.step()
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, others))
    opt.zero_grad(set_to_none=True)
    fired = []
    h = m.classification_head.register_forward_hook(lambda mod, i, o: fired.append(1))
    assert not m._head_phase_fusable(torch.zeros(4, 192, device=dev()))
    h.remove()
    assert m._head_phase_fusable(torch.zeros(4, 192, device=dev()))


def test_parameter_gradients_on_their_own_stream_equal_the_inline_launch():
    """The head / KAN parameter gradients are computed on a side stream and joined by an autograd final callback: over several steps
    (optimizer included) every parameter stays bit-identical to a run with the launch inline on the backward's stream."""
    from models.rovit_kan import RoViTKAN
    from rovit_hip.functions import HeadPhaseFn
    from rovit_hip.losses import JointLoss
    from rovit_hip.optim import RoViTAdamW
    x = torch.randn(8, 3, 224, 224, device=dev())
    y = torch.randint(0, 4, (8,), device=dev())
    finals = []
    for side in (True, False):
        torch.manual_seed(5)
        m = RoViTKAN(pretrained=False, dropout=0.3).to(dev()).train()
        opt = RoViTAdamW(m, lr=1e-3)
        loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0)
        orig = m._forward_head_phase
        if not side:
            import functools
            real_apply = HeadPhaseFn.apply

            def no_side(features, cfg, *params, _real=real_apply):
                return _real(features, dict(cfg, dw_side_stream=False), *params)
            HeadPhaseFn.apply = staticmethod(no_side)
        try:
            for _ in range(4):
                opt.zero_grad(set_to_none=True)
                loss_fn(m(x), y, y, 4)['total_loss'].backward()
                assert (not side) or HeadPhaseFn._pending.get(0) is None        # the final callback consumed the event
                opt.step()
        finally:
            if not side:
                HeadPhaseFn.apply = real_apply
        torch.cuda.synchronize()
        finals.append({n: p.detach().clone() for n, p in m.named_parameters()})
    for n in finals[0]:
        assert torch.equal(finals[0][n], finals[1][n]), n


def test_joint_loss_takes_int64_severity_labels_without_a_cast_launch():
    from rovit_hip.losses import JointLoss
    g = torch.Generator().manual_seed(3)
    out = {'cls_logits': torch.randn(19, 4, generator=g).to(dev()).requires_grad_(True),
           'ordinal_logits': torch.randn(19, 3, generator=g).to(dev()).requires_grad_(True),
           'mu': torch.randn(19, 1, generator=g).to(dev()).requires_grad_(True), 'log_var': torch.randn(19, 1, generator=g).to(dev()).requires_grad_(True),
           'kan_severity': (3 * torch.rand(19, 1, generator=g)).to(dev()).requires_grad_(True)}
    y = torch.randint(0, 4, (19,), generator=g).to(dev())
    loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0)
    a = loss_fn(out, y, y, 4)
    a['total_loss'].backward()
    ga = {k: v.grad.clone() for k, v in out.items()}
    for v in out.values():
        v.grad = None
    b = loss_fn(out, y, y.float(), 4)
    b['total_loss'].backward()
    for k in ('cls_loss', 'ord_loss', 'unc_loss', 'kan_loss', 'total_loss'):
        assert torch.equal(a[k], b[k]), k
    for k, v in out.items():
        assert torch.equal(ga[k], v.grad), k
