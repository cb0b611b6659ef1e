"""CPU tests: the oracle (oracle/ref_cpu.py) against the golden vectors generated from the
reference's own classes (oracle/make_golden.py) and against its published known answers."""
import hashlib
import os

import numpy as np
import torch

from oracle import ref_cpu

T = torch.from_numpy


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def sd_of(g, prefix='sd.'):
    return {k[len(prefix):]: T(g[k]) for k in g.files if k.startswith(prefix)}


def test_basis_matches_reference_table(golden_dir):
    g = load(golden_dir, 'kan_basis')
    for G in (5, 32):
        x, knots, ref = T(g[f'g{G}.x']), T(g[f'g{G}.knots']), T(g[f'g{G}.basis'])
        got = ref_cpu.truncated_bspline_basis(x, knots, 3)
        assert got.shape == ref.shape
        assert torch.equal(got, ref), float((got - ref).abs().max())     # same op order => bit-exact
        # the truncation: identically zero at and beyond knots[num_basis]  
        nb = knots.numel() - 4
        beyond = x >= knots[nb]
        assert beyond.any() and float(ref[beyond].abs().max()) == 0.0


def test_closed_form_matches_recursion(golden_dir):
    g = load(golden_dir, 'kan_basis')
    for G in (5, 32):
        x, knots, ref = T(g[f'g{G}.x']), T(g[f'g{G}.knots']), T(g[f'g{G}.basis'])
        j, vals = ref_cpu.closed_form_basis(x, knots)
        nb = knots.numel() - 4
        dense = torch.zeros_like(ref)
        for m in range(4):
            idx = j - m
            ok = (idx >= 0) & (idx < nb)
            dense.scatter_add_(-1, idx.clamp(0, nb - 1).unsqueeze(-1), (vals[..., m] * ok).unsqueeze(-1))
        assert float((dense - ref).abs().max()) < 2e-6


def _check_kan(golden_dir, name):
    g = load(golden_dir, name)
    sd = sd_of(g)
    deg = int(g['degree'])
    x = T(g['x']).clone().requires_grad_(True)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    y = ref_cpu.kan_module_forward(x, sd, degree=deg)
    assert float((y - T(g['y'])).abs().max()) < 2e-5
    assert float(y.min()) >= 0.0 and float(y.max()) <= 3.0
    (y * T(g['w'])).sum().backward()
    assert float((x.grad - T(g['dx'])).abs().max()) < 5e-5 * max(1.0, float(T(g['dx']).abs().max()))
    for k, v in sd.items():
        if 'knots' in k:
            continue
        ref = T(g['grad.' + k])
        assert float((v.grad - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max())), k
    yl = ref_cpu.kan_module_forward(T(g['x']), {k: v.detach() for k, v in sd.items()}, degree=deg, loop=True)
    assert float((yl - T(g['y'])).abs().max()) < 2e-6


def test_kan_mini(golden_dir):
    _check_kan(golden_dir, 'kan_mini')


def test_kan_default(golden_dir):
    _check_kan(golden_dir, 'kan_default')
    g = load(golden_dir, 'kan_default')
    n = sum(v.size for k, v in g.items() if k.startswith('sd.') and 'knots' not in k)
    assert n == 106705            # published KAN parameter count


def test_kan_g32(golden_dir):
    _check_kan(golden_dir, 'kan_g32')


def test_kan_init_shapes():
    sd = ref_cpu.init_kan_state([192, 64, 16, 1], 32, 3)
    assert sum(v.numel() for k, v in sd.items() if 'knots' not in k) == 466561
    assert sd['kan_layers.0.knots'].numel() == 38


def test_heads(golden_dir):
    g = load(golden_dir, 'heads')
    sd = sd_of(g)
    x = T(g['x']).clone().requires_grad_(True)
    for v in sd.values():
        v.requires_grad_(True)
    out = ref_cpu.heads_forward(x, sd, 4)
    for k in ('cls_logits', 'ordinal_logits', 'mu', 'log_var'):
        assert float((out[k] - T(g[k])).abs().max()) < 1e-5, k
    assert float(out['log_var'].abs().max()) <= 10.0
    assert float((ref_cpu.ordinal_probabilities(out['ordinal_logits']) - T(g['ord_probs'])).abs().max()) < 1e-6
    assert float((ref_cpu.ordinal_severity(out['ordinal_logits']) - T(g['ord_severity'])).abs().max()) < 1e-6
    loss = sum((out[k] * T(g[w])).sum() for k, w in (('cls_logits', 'w.cls'), ('ordinal_logits', 'w.ord'),
                                                      ('mu', 'w.mu'), ('log_var', 'w.lv')))
    loss.backward()
    assert float((x.grad - T(g['dx'])).abs().max()) < 1e-5
    for k, v in sd.items():
        assert float((v.grad - T(g['grad.' + k])).abs().max()) < 2e-5, k
    counts = {h: sum(v.numel() for k, v in sd.items() if k.startswith(h)) for h in
              ('classification_head', 'ordinal_head', 'uncertainty_head')}
    assert counts == {'classification_head': 25220, 'ordinal_head': 25091, 'uncertainty_head': 24962}


def test_stage_gating_keys():
    sd = ref_cpu.init_heads_state()
    f = torch.randn(3, 192)
    for stage, none_keys in ((1, {'ordinal_logits', 'mu', 'log_var'}), (2, {'mu', 'log_var'}), (3, set()), (4, set())):
        out = ref_cpu.heads_forward(f, sd, stage)
        assert {k for k, v in out.items() if v is None} == none_keys


def test_joint_loss(golden_dir):
    g = load(golden_dir, 'joint_loss')
    y, alpha = T(g['y']), T(g['alpha'])
    for stage in (1, 2, 3, 4):
        outd = {k[3:]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith('in.')}
        l = ref_cpu.joint_loss(outd, y, y, stage, alpha=alpha)
        for k in ('cls_loss', 'ord_loss', 'unc_loss', 'kan_loss', 'total_loss'):
            assert abs(float(l[k]) - float(g[f's{stage}.{k}'])) < 1e-5, (stage, k)
        l['total_loss'].backward()
        for k, v in outd.items():
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            assert float((got - T(g[f's{stage}.grad.{k}'])).abs().max()) < 1e-6, (stage, k)


def _check_vit(golden_dir, name):
    g = load(golden_dir, name)
    depth, batch, seed = int(g['depth']), int(g['batch']), int(g['seed'])
    gen = torch.Generator().manual_seed(seed)
    sd = ref_cpu.init_vit_state(depth, gen)
    flat = torch.cat([sd[k].flatten() for k in sorted(sd)])
    assert flat.numel() == int(g['n_params'])
    x = torch.randn(batch, 3, 224, 224, generator=gen)
    if hashlib.sha256(flat.numpy().tobytes()).hexdigest() != str(g['weights_sha256']):
        import pytest
        pytest.skip('torch CPU generator stream differs from the one the fixture was made with')
    assert np.array_equal(x[0, :, :2, :4].numpy(), g['x_probe'])
    with torch.no_grad():
        f = ref_cpu.vit_forward(x, sd)
    assert f.shape == (batch, 192)
    assert float((f - T(g['features'])).abs().max()) < 2e-5      # vs transformers.ViTModel


def test_vit_depth2_vs_hf(golden_dir):
    _check_vit(golden_dir, 'vit_depth2')


def test_vit_depth12_vs_hf(golden_dir):
    _check_vit(golden_dir, 'vit_depth12')


def test_param_counts_match_published():
    """Known answers the reference publishes (outputs/ablation/full_model/test_metrics.json:11)."""
    shapes = ref_cpu.vit_param_shapes()
    n_backbone = sum(int(np.prod(s)) for s in shapes.values())
    assert n_backbone == 5524416
    sd = ref_cpu.init_rovit_state(depth=1)
    n_rest = sum(v.numel() for k, v in sd.items() if not k.startswith('backbone.') and 'knots' not in k)
    assert n_backbone + n_rest == 5706394


def test_config1_plumbing_cpu():
    """BASELINE.json configs[0]: full model forward on 8 random 224x224 images, CPU path, logit shape."""
    torch.manual_seed(0)
    sd = ref_cpu.init_rovit_state(depth=12, seed=0)
    x = torch.randn(8, 3, 224, 224)
    with torch.no_grad():
        out = ref_cpu.rovit_forward(x, sd, stage=4)
    assert out['cls_logits'].shape == (8, 4) and out['features'].shape == (8, 192)
    assert out['ordinal_logits'].shape == (8, 3) and out['mu'].shape == (8, 1) and out['log_var'].shape == (8, 1)
    assert out['kan_severity'].shape == (8, 1)
    assert float(out['kan_severity'].min()) >= 0 and float(out['kan_severity'].max()) <= 3


def test_philox_restatement_against_the_published_known_answer_vectors():
    """oracle/philox.py (the checker of the dropout masks the fused head-phase kernel draws) against the known-answer vectors of the
    Random123 distribution for philox4x32 with 10 rounds (Salmon et al., SC'11: kat_vectors), and the mask helper's shape / scaling."""
    import numpy as np
    from oracle.philox import head_phase_masks, philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in philox4x32_10(ctr, key)) == want
    masks = head_phase_masks(64, 128, 0.3, 1234, 40)
    assert len(masks) == 3 and all(m.shape == (64, 128) and m.dtype == np.float32 for m in masks)
    for m in masks:
        vals = np.unique(m)
        assert set(vals.tolist()) <= {0.0, float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.3)))}
        assert abs(float((m > 0).mean()) - 0.7) < 0.03
    assert not np.array_equal(masks[0], masks[1])
    assert not np.array_equal(masks[0], head_phase_masks(64, 128, 0.3, 1234, 44)[0])          # another offset, another draw
