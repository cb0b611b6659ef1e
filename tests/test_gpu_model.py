"""GPU parity tests of the composed path (backbone, full RoViT-KAN, loss backward) against the CPU oracle.

Tolerances.  The HIP backbone computes its GEMMs/attention with bf16 operands and fp32 accumulation (residual
stream, LayerNorm statistics, softmax, heads and KAN stay fp32), i.e. the precision class of the reference's own
CUDA path (fp16 autocast, training/trainer.py:99-129).  Against the fp32 CPU oracle the stated bf16 tolerance is
5e-2 max-abs and 1.2e-2 RMS on the LayerNorm'd features / logits (values are O(1); measured max 1.1e-2 .. 3.4e-2,
RMS ~5e-3 over seeds) (1e-3 applies to the fp32 heads/KAN given identical features, see
test_gpu_kernels.py), and the class argmax must agree wherever the oracle's top-2 margin exceeds that tolerance.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu  # noqa: E402  (checker only)

BF16_TOL = 3e-2          # ~1.5x the worst case measured on MI355X over rounds 1-2 (max-abs 1.3e-2 .. 2.1e-2)
BF16_RMS = 8e-3          # measured RMS 4.3e-3 .. 6.1e-3


def dev():
    return torch.device('cuda:0')


@pytest.fixture(params=['fused_mlp_half', 'two_launch_mlp_half'])
def mlp_half(request):
    """rovit_vit_forward / _backward pick the MLP-half kernels by batch size (one fused launch from 34 000 token rows, two launches
    below): the small parity cases run under BOTH so that the path of the benchmark batch is checked against the oracle too."""
    from rovit_hip import native
    from rovit_hip.functions import VitEngine
    # (round 4: `mlp_path` is an argument of the forward / backward calls, no longer a library-wide setter)
    VitEngine.default_mlp_path = native.MLP_ONE_LAUNCH if request.param == 'fused_mlp_half' else native.MLP_TWO_LAUNCH
    yield request.param
    VitEngine.default_mlp_path = native.MLP_AUTO


def _vit(depth, sd):
    from models.backbone import DeiTTiny
    m = DeiTTiny(depth)
    m.load_state_dict(sd)
    return m.to(dev())


@pytest.mark.parametrize('name', ['vit_depth2', 'vit_depth12'])
def test_backbone_forward_vs_hf_golden(golden_dir, name, mlp_half):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    depth, batch, seed = int(g['depth']), int(g['batch']), int(g['seed'])
    gen = torch.Generator().manual_seed(seed)
    sd = ref_cpu.init_vit_state(depth, gen)
    x = torch.randn(batch, 3, 224, 224, generator=gen)
    with torch.no_grad():
        ref = ref_cpu.vit_forward(x, sd)
    m = _vit(depth, sd)
    with torch.no_grad():
        f = m(x.to(dev())).cpu()
    assert f.shape == (batch, 192)
    err = float((f - ref).abs().max())
    rms = float((f - ref).pow(2).mean().sqrt())
    print(name, 'max|features - oracle| =', err, 'rms', rms)
    assert err < BF16_TOL and rms < BF16_RMS
    # the committed HF-ViT features are only comparable on the fixture's own input: the probe pins the generator stream
    assert np.array_equal(x[0, :, :2, :4].numpy(), g['x_probe']), 'generator stream differs from the one the fixture was made with'
    assert float((f - torch.from_numpy(g['features'])).abs().max()) < BF16_TOL


def test_backbone_backward_vs_oracle(mlp_half):
    depth, B = 2, 3
    gen = torch.Generator().manual_seed(21)
    sd = ref_cpu.init_vit_state(depth, gen)
    x = torch.randn(B, 3, 224, 224, generator=gen)
    w = torch.randn(B, 192, generator=gen)
    ref_p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (ref_cpu.vit_forward(x, ref_p) * w).sum().backward()
    m = _vit(depth, sd)
    f = m(x.to(dev()))
    (f * w.to(dev())).sum().backward()
    worst = 0.0
    for k, p in m.named_parameters():
        ref = ref_p[k].grad
        got = p.grad.cpu()
        assert got.shape == ref.shape, k
        rel = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-8))
        cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
        worst = max(worst, rel)
        assert cos > 0.999, (k, cos, rel)
        assert rel < 6e-2, (k, rel)
    print('worst relative grad error', worst)


def test_backbone_grad_accumulation_and_frozen():
    depth, B = 1, 2
    gen = torch.Generator().manual_seed(5)
    sd = ref_cpu.init_vit_state(depth, gen)
    x = torch.randn(B, 3, 224, 224, generator=gen).to(dev())
    m = _vit(depth, sd)
    m(x).sum().backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m(x).sum().backward()                                  # second backward accumulates
    for k, p in m.named_parameters():
        assert float((p.grad - 2 * g1[k]).abs().max()) <= 1e-5 * float(g1[k].abs().max() + 1e-6), k
    for p in m.parameters():
        p.grad = None
        p.requires_grad = False
    f = m(x)
    assert not f.requires_grad


def _full_model(sd):
    from models.rovit_kan import RoViTKAN
    m = RoViTKAN(pretrained=False)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.to(dev())


def test_full_model_forward_all_stages_vs_oracle():
    sd = ref_cpu.init_rovit_state(seed=3)
    torch.manual_seed(0)
    x = torch.randn(8, 3, 224, 224)
    m = _full_model(sd).eval()
    for stage in (1, 2, 3, 4):
        m.curriculum_stage = stage
        with torch.no_grad():
            out = m(x.to(dev()))
            ref = ref_cpu.rovit_forward(x, sd, stage)
        assert set(out.keys()) == {'cls_logits', 'features', 'ordinal_logits', 'mu', 'log_var', 'kan_severity'}
        for k, r in ref.items():
            if r is None:
                assert out[k] is None, (stage, k)
                continue
            err = float((out[k].cpu() - r).abs().max())
            assert out[k].shape == r.shape
            if k == 'kan_severity':
                # The reference's truncated spline jumps to zero at x = atanh(knots[num_basis]),
                # so features that differ by bf16 rounding flip a few basis terms and move the severity by
                # O(0.1): end-to-end it is only comparable on identical features (checked to 1e-3 below).
                continue
            assert err < BF16_TOL, (stage, k, err)
            assert float((out[k].cpu() - r).pow(2).mean().sqrt()) < BF16_RMS, (stage, k)
        # class argmax: identical wherever the oracle's top-2 margin exceeds twice the MEASURED logit error of this run
        # (the stated 5e-2 bound would exclude every sample of a random-init model, whose margins are ~1e-2: the
        # comparison must not be vacuous); excluded samples are counted
        lerr = float((out['cls_logits'].cpu() - ref['cls_logits']).abs().max())
        top2 = ref['cls_logits'].topk(2, dim=1).values
        decided = (top2[:, 0] - top2[:, 1]) > 2 * lerr
        agree = int((out['cls_logits'].cpu().argmax(1) == ref['cls_logits'].argmax(1)).sum())
        print(f'stage {stage}: class argmax compared on {int(decided.sum())} of {decided.numel()} samples (oracle top-2 margin > '
              f'2 x {lerr:.4f}), {int((~decided).sum())} excluded; agreement on all samples {agree}/{decided.numel()}')
        assert int(decided.sum()) >= decided.numel() // 2, 'argmax comparison is vacuous'
        assert torch.equal(out['cls_logits'].cpu().argmax(1)[decided], ref['cls_logits'].argmax(1)[decided])
        # fp32 heads / KAN given the SAME features: north_star's 1e-3
        f = out['features'].cpu()
        hr = ref_cpu.heads_forward(f, sd, stage)
        for k in ('cls_logits', 'ordinal_logits', 'mu', 'log_var'):
            if hr[k] is not None:
                assert float((out[k].cpu() - hr[k]).abs().max()) < 1e-3, (stage, k)
        if stage == 4:
            kr = ref_cpu.kan_module_forward(f, sd, 'kan_module.')
            assert float((out['kan_severity'].cpu() - kr).abs().max()) < 1e-3
            assert float(out['kan_severity'].min()) >= 0 and float(out['kan_severity'].max()) <= 3
    with pytest.raises(AssertionError):
        m.curriculum_stage = 5


def test_full_model_joint_loss_backward_vs_oracle(mlp_half):
    sd = ref_cpu.init_rovit_state(depth=12, seed=4)
    torch.manual_seed(1)
    B = 4
    x = torch.randn(B, 3, 224, 224)
    y = torch.randint(0, 4, (B,))
    ref_p = {k: (v.clone().requires_grad_(True) if 'knots' not in k else v) for k, v in sd.items()}
    m = _full_model(sd).eval()        # eval: dropout off, as in the oracle
    out = m(x.to(dev()))
    yd = y.to(dev())
    # The KAN spline is discontinuous at x = atanh(knots[num_basis]): a feature that crosses the
    # cutoff under bf16 rounding changes the severity AND its gradient by O(1).  To compare the backward of the
    # two implementations (not the position of a discontinuity) the oracle's heads/KAN are evaluated at the SAME
    # feature values the HIP path produced, while its gradient still flows through its own fp32 backbone.
    f_ref = ref_cpu.vit_forward(x, ref_p, prefix='backbone.model.')
    f_used = f_ref + (out['features'].detach().cpu() - f_ref).detach()
    ro = ref_cpu.heads_forward(f_used, ref_p, 4)
    ro['kan_severity'] = ref_cpu.kan_module_forward(f_used, ref_p, 'kan_module.')
    rl = ref_cpu.joint_loss(ro, y, y, 4)
    rl['total_loss'].backward()
    # the loss itself is O(B) plain torch on device (row f-1 of the scope table, not yet a HIP kernel)
    gl = ref_cpu.joint_loss(out, yd, yd, 4, alpha=None)
    assert abs(float(gl['total_loss']) - float(rl['total_loss'])) < BF16_TOL
    gl['total_loss'].backward()
    bad = []
    for k, p in m.named_parameters():
        ref = ref_p[k].grad
        got = p.grad.cpu()
        rel = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-8))
        cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
        if cos < 0.995 or rel > 0.1:
            bad.append((k, round(cos, 5), round(rel, 4)))
    assert not bad, bad[:10]


def test_train_mode_dropout_runs_and_stage_gating_backward():
    sd = ref_cpu.init_rovit_state(depth=12, seed=6)
    m = _full_model(sd).train()
    x = torch.randn(2, 3, 224, 224, device=dev())
    for stage in (1, 2, 3, 4):
        m.curriculum_stage = stage
        for p in m.parameters():
            p.grad = None
        out = m(x)
        y = torch.tensor([1, 3], device=dev())
        ref_cpu.joint_loss(out, y, y, stage)['total_loss'].backward()
        has = {n.split('.')[0] for n, p in m.named_parameters() if p.grad is not None and float(p.grad.abs().sum()) > 0}
        expect = {'backbone', 'classification_head'} | ({'ordinal_head'} if stage >= 2 else set()) | \
                 ({'uncertainty_head'} if stage >= 3 else set()) | ({'kan_module'} if stage >= 4 else set())
        assert has == expect, (stage, has)


def test_flat_adamw_matches_torch_clip_and_adamw():
    """rovit_hip.optim.RoViTAdamW == clip_grad_norm_(1.0) + torch AdamW with the reference's two lr groups."""
    import copy
    from models.rovit_kan import RoViTKAN
    from rovit_hip.optim import RoViTAdamW
    from rovit_hip.losses import JointLoss
    torch.manual_seed(0)
    m1 = RoViTKAN(pretrained=False).to(dev()).eval()
    m2 = copy.deepcopy(m1)
    bb = [p for n, p in m2.named_parameters() if 'backbone' in n]
    hd = [p for n, p in m2.named_parameters() if 'backbone' not in n]
    ref_opt = torch.optim.AdamW([{'params': bb, 'lr': 1e-3 / 10}, {'params': hd, 'lr': 1e-3}], weight_decay=1e-2)
    opt = RoViTAdamW(m1, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    x = torch.randn(2, 3, 224, 224, device=dev())
    y = torch.tensor([0, 2], device=dev())
    lf = JointLoss()
    # ONE step from identical states: with bf16 GEMM operands the forward is a discontinuous function of the
    # parameters at the 1e-2 level, so two optimizers that agree to 1e-8 still drift apart over several steps.
    for m, o in ((m1, opt), (m2, ref_opt)):
        o.zero_grad(set_to_none=True)
        lf(m(x), y, y, 4)['total_loss'].backward()
    before = {n: p.detach().clone() for n, p in m2.named_parameters()}
    gn_ref = torch.nn.utils.clip_grad_norm_(m2.parameters(), 1.0)
    ref_opt.step()
    opt.step()
    assert abs(float(opt.last_grad_norm) - float(gn_ref)) < 1e-4 * float(gn_ref)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert n1 == n2
        upd = float((p2.detach() - before[n2]).abs().max())
        assert upd > 0, n1
        assert float((p1.detach() - p2.detach()).abs().max()) < 2e-3 * upd + 1e-9, n1


@pytest.mark.parametrize('B', [1, 3, 17])
def test_inference_workspace_equals_training_forward_ragged_batches(B):
    """no_grad forward (activation buffers shared by all blocks) == grad-mode forward (per-block buffers), for batch
    sizes whose row counts are not multiples of any tile size (M = 197, 591, 3349)."""
    sd = ref_cpu.init_vit_state(12, torch.Generator().manual_seed(8))
    m = _vit(12, sd)
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(B)).to(dev())
    with torch.no_grad():
        f_inf = m(x).clone()
    f_trn = m(x)
    assert f_trn.requires_grad and not f_inf.requires_grad
    assert torch.equal(f_inf, f_trn.detach())
    with torch.no_grad():
        ref = ref_cpu.vit_forward(x.cpu(), sd)
    assert float((f_inf.cpu() - ref).abs().max()) < BF16_TOL
    f_trn.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_forward_is_deterministic_and_batch_independent():
    """same image -> same features whatever the batch it sits in (no cross-sample coupling, no atomics in forward)."""
    sd = ref_cpu.init_vit_state(12, torch.Generator().manual_seed(9))
    m = _vit(12, sd)
    x = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev())
    with torch.no_grad():
        a = m(x).clone()
        b = m(x).clone()
        c = m(x[2:4]).clone()
    assert torch.equal(a, b)
    assert torch.equal(a[2:4], c)


def test_backward_is_deterministic():
    sd = ref_cpu.init_vit_state(2, torch.Generator().manual_seed(10))
    m = _vit(2, sd)
    x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev())
    grads = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        m(x).square().sum().backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    for g0, g1, (n, _) in zip(grads[0], grads[1], m.named_parameters()):
        assert torch.equal(g0, g1), n            # fixed-order reductions everywhere: bitwise, LayerNorm affines included


def test_attention_output_taps_vs_oracle():
    """DeiTTinyBackbone.get_attention_maps (reference models/backbone.py:37-62): per-block attention-module outputs.
    The oracle collects the same tensors; tolerance is the bf16 one relative to each tap's scale (taps are stored in
    bf16 and are O(0.1) at init scale)."""
    from models.backbone import DeiTTinyBackbone
    depth, B = 12, 3
    gen = torch.Generator().manual_seed(31)
    sd = ref_cpu.init_vit_state(depth, gen)
    x = torch.randn(B, 3, 224, 224, generator=gen)
    ref_taps = []
    with torch.no_grad():
        ref_f = ref_cpu.vit_forward(x, sd, attn_taps=ref_taps)
    bb = DeiTTinyBackbone(pretrained=False)
    bb.model.load_state_dict(sd)
    bb = bb.to(dev())
    with torch.no_grad():
        f_plain = bb(x.to(dev()))
    taps = bb.get_attention_maps(x.to(dev()))
    assert len(taps) == depth and all(t.shape == (B, 197, 192) and t.dtype == torch.float32 for t in taps)
    for i, (t, r) in enumerate(zip(taps, ref_taps)):
        scale = float(r.abs().max())
        err = float((t.cpu() - r).abs().max())
        assert err < BF16_TOL * max(scale, 1.0) and err < 0.1 * scale + 1e-3, (i, err, scale)
    with torch.no_grad():                                             # tapping does not disturb the plain forward
        assert torch.equal(bb(x.to(dev())), f_plain)
    assert float((f_plain.cpu() - ref_f).abs().max()) < BF16_TOL


def test_two_stream_schedule_vs_oracle_and_on_a_side_stream(mlp_half):
    """B >= 16 takes the two-stream schedule (half-batch forward chains, dgrad/wgrad backward streams): depth 4 so
    that every rotating hand-over buffer of the backward is reused at least once.  Checked against the oracle, and
    the same step launched from a non-default torch stream must reproduce it bit for bit (the default stream is the
    NULL stream, a separate code path for event waits)."""
    depth, B = 4, 18
    gen = torch.Generator().manual_seed(77)
    sd = ref_cpu.init_vit_state(depth, gen)
    x = torch.randn(B, 3, 224, 224, generator=gen)
    w = torch.randn(B, 192, generator=gen)
    ref_p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_f = ref_cpu.vit_forward(x, ref_p)
    (ref_f * w).sum().backward()
    m = _vit(depth, sd)
    xd, wd = x.to(dev()), w.to(dev())
    f = m(xd)
    (f * wd).sum().backward()
    torch.cuda.synchronize()
    err = (f.detach().cpu() - ref_f.detach()).abs().amax(1)
    assert float(err.max()) < BF16_TOL, err                      # every sample, both halves
    g0 = {}
    for k, p in m.named_parameters():
        ref, got = ref_p[k].grad, p.grad.cpu()
        rel = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-8))
        cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
        assert cos > 0.999 and rel < 6e-2, (k, cos, rel)
        g0[k] = p.grad.clone()
        p.grad = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f2 = m(xd)
        (f2 * wd).sum().backward()
    side.synchronize()
    assert torch.equal(f2.detach(), f.detach())
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, g0[k]), k


def test_bucketed_grad_sync_one_rank_matches_plain_backward(mlp_half):
    """GradSync with the RCCL process group of ONE rank (collectives forced): block-range backward with the deferred
    join (rovit_vit_backward_notify) + all-reduce on the side stream must leave exactly the gradients of the plain
    backward (AVG over one rank is the identity)."""
    import torch.distributed as dist
    from models.rovit_kan import RoViTKAN
    from rovit_hip.losses import JointLoss
    from rovit_hip.parallel import GradSync
    import socket
    with socket.socket() as sk:                                  # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.manual_seed(3)
    m = RoViTKAN(pretrained=False).to(dev()).eval()
    x = torch.randn(20, 3, 224, 224, device=dev())
    y = torch.randint(0, 4, (20,), device=dev())
    lf = JointLoss()

    def grads():
        for p in m.parameters():
            p.grad = None
        lf(m(x), y, y, 4)['total_loss'].backward()
        return {n: p.grad.clone() for n, p in m.named_parameters()}
    g_plain = grads()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        sync = GradSync(m, buckets=3, force=True)
        assert sync.active and m.backbone.model.engine.notify_stream is not None
        for p in m.parameters():
            p.grad = None
        lf(m(x), y, y, 4)['total_loss'].backward()
        sync.finish()
        torch.cuda.synchronize()
        assert len(sync.reducer.issued) == 4                       # 3 backbone buckets + heads
        for n, p in m.named_parameters():
            assert torch.equal(p.grad, g_plain[n]), n
        # with the optimizer attached the head / KAN bucket is the optimizer's own flat gradient buffer, reduced in place:
        # param.grad become views of it, the values stay those of the plain backward, and step() does not pack again
        from rovit_hip.optim import RoViTAdamW
        opt = RoViTAdamW(m, lr=1e-4)
        sync2 = GradSync(m, buckets=2, force=True, optimizer=opt)
        for p in m.parameters():
            p.grad = None
        lf(m(x), y, y, 4)['total_loss'].backward()
        sync2.finish()
        torch.cuda.synchronize()
        assert len(sync2.reducer.issued) == 3                      # 2 backbone buckets + one contiguous head/KAN run
        packed = opt.grad_view_ptrs()
        for n, p in m.named_parameters():
            assert torch.equal(p.grad, g_plain[n]), n
            if not n.startswith('backbone.'):
                assert p.grad.data_ptr() in packed, n
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        opt.step()
        assert any(not torch.equal(before[n], p.detach()) for n, p in m.named_parameters())
    finally:
        eng = m.backbone.model.engine
        eng.backward_ranges = eng.range_hook = eng.notify_stream = None
        if created:
            dist.destroy_process_group()


def test_training_loop_reduces_loss_and_respects_stage_gating():
    """A short real training loop on one fixed batch (RoViTAdamW + fused JointLoss + HIP forward/backward): the loss
    goes down, parameters of heads that are inactive at the current curriculum stage are not touched (torch.optim
    skips parameters without gradients; so does the flat optimizer, per module segment), and they start moving once
    their stage is reached."""
    from models.rovit_kan import RoViTKAN
    from rovit_hip.optim import RoViTAdamW
    from rovit_hip.losses import JointLoss
    torch.manual_seed(11)
    m = RoViTKAN(pretrained=False, dropout=0.0).to(dev()).train()
    opt = RoViTAdamW(m, lr=2e-3, weight_decay=1e-4, max_grad_norm=1.0)
    lf = JointLoss()
    x = torch.randn(32, 3, 224, 224, device=dev())
    y = torch.randint(0, 4, (32,), device=dev())

    def snapshot():
        return {n: p.detach().clone() for n, p in m.named_parameters()}

    def run(stage, steps):
        m.curriculum_stage = stage
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            loss = lf(m(x), y, y, stage)['total_loss']
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses
    before = snapshot()
    l1 = run(1, 12)
    after1 = snapshot()
    assert l1[-1] < 0.8 * l1[0], l1
    moved = {n.split('.')[0] for n in before if not torch.equal(before[n], after1[n])}
    assert moved == {'backbone', 'classification_head'}, moved
    l4 = run(4, 12)
    after4 = snapshot()
    assert l4[-1] < 0.8 * l4[0], l4
    moved = {n.split('.')[0] for n in before if not torch.equal(after1[n], after4[n])}
    assert moved == {'backbone', 'classification_head', 'ordinal_head', 'uncertainty_head', 'kan_module'}, moved
    assert all(torch.isfinite(p).all() for p in m.parameters())
    seg_t = {s.name: s.t for s in opt.segments}
    assert seg_t['classification_head'] == 24 and seg_t['kan_module'] == 12 and opt.t == 24, seg_t


def test_optimizer_is_a_torch_optimizer_scheduler_and_state_dict_round_trip():
    """CosineAnnealingLR / get_lr of the reference (training/optimizer.py:35-49) drive RoViTAdamW's two parameter
    groups; state_dict round-trips the flat moments so a resumed optimizer takes the identical next step."""
    import copy
    from types import SimpleNamespace
    from models.rovit_kan import RoViTKAN
    from rovit_hip.optim import RoViTAdamW, build_optimizer, build_scheduler, get_lr
    from rovit_hip.losses import JointLoss
    torch.manual_seed(5)
    cfg = SimpleNamespace(train=SimpleNamespace(learning_rate=1e-3, weight_decay=1e-4, epochs=10),
                          flags=SimpleNamespace(gradient_clip=1.0))
    m = RoViTKAN(pretrained=False, dropout=0.0).to(dev()).train()
    opt = build_optimizer(m, cfg)
    assert isinstance(opt, torch.optim.Optimizer) and isinstance(opt, RoViTAdamW)
    assert [g['lr'] for g in opt.param_groups] == [1e-4, 1e-3] and get_lr(opt) == 1e-4
    sched = build_scheduler(opt, cfg)
    x = torch.randn(4, 3, 224, 224, device=dev())
    y = torch.randint(0, 4, (4,), device=dev())
    lf = JointLoss()

    def one_step(model, optimizer):
        optimizer.zero_grad()
        lf(model(x), y, y, 4)['total_loss'].backward()
        optimizer.step()
    one_step(m, opt)
    sched.step()
    lrs = [g['lr'] for g in opt.param_groups]
    assert lrs[0] < 1e-4 and lrs[1] < 1e-3 and abs(lrs[1] / lrs[0] - 10.0) < 0.2       # cosine decay of both groups
    # resume: clone model + optimizer state, both take one more step -> identical parameters
    m2 = copy.deepcopy(m)
    opt2 = build_optimizer(m2, cfg)
    opt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    assert [g['lr'] for g in opt2.param_groups] == lrs and opt2.t == opt.t
    one_step(m, opt)
    one_step(m2, opt2)
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * float(p.abs().max() + 1e-12), n


def test_attention_probability_taps_vs_oracle():
    """Softmax probabilities per block (B,3,197,197): rows sum to 1, and match the oracle within the bf16 tolerance of
    the q/k operands (probabilities are <= 1; measured max |diff| ~1e-3 at init scale)."""
    from models.backbone import DeiTTinyBackbone
    depth, B = 3, 2
    gen = torch.Generator().manual_seed(41)
    sd = ref_cpu.init_vit_state(depth, gen)
    x = torch.randn(B, 3, 224, 224, generator=gen)
    ref_p = []
    with torch.no_grad():
        ref_cpu.vit_forward(x, sd, attn_probs=ref_p)
    bb = DeiTTinyBackbone(pretrained=False)
    bb.model = bb.model.__class__(depth)
    bb.model.load_state_dict(sd)
    bb = bb.to(dev())
    probs = bb.get_attention_probabilities(x.to(dev()))
    assert len(probs) == depth
    for i, (p, r) in enumerate(zip(probs, ref_p)):
        assert p.shape == (B, 3, 197, 197) and p.dtype == torch.float32
        assert float((p.sum(-1) - 1).abs().max()) < 1e-5
        assert float((p.cpu() - r).abs().max()) < 5e-3, i


def test_frozen_backbone_training_step_and_checkpoint_round_trip():
    """Curriculum epochs 1-5 of the reference train with the backbone frozen (trainer.py:63,245): the backbone's backward
    is skipped entirely (no gradients, parameters untouched), heads still train; after unfreeze_backbone() everything
    moves.  state_dict() keys are the reference's and a save/load round trip reproduces the outputs bit for bit, with
    the parameters living as views of the optimizer's flat buffers."""
    import io
    from models.rovit_kan import RoViTKAN
    from rovit_hip.optim import RoViTAdamW
    from rovit_hip.losses import JointLoss
    torch.manual_seed(17)
    m = RoViTKAN(pretrained=False, dropout=0.0).to(dev()).train()
    opt = RoViTAdamW(m, lr=1e-3)
    lf = JointLoss()
    x = torch.randn(6, 3, 224, 224, device=dev())
    y = torch.randint(0, 4, (6,), device=dev())
    m.freeze_backbone()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.zero_grad()
    lf(m(x), y, y, 4)['total_loss'].backward()
    assert all(p.grad is None for n, p in m.named_parameters() if n.startswith('backbone.'))
    opt.step()
    for n, p in m.named_parameters():
        assert torch.equal(p, before[n]) == n.startswith('backbone.'), n
    m.unfreeze_backbone()
    opt.zero_grad()
    lf(m(x), y, y, 4)['total_loss'].backward()
    opt.step()
    assert not torch.equal(m.backbone.model.blocks[0].attn.qkv.weight, before['backbone.model.blocks.0.attn.qkv.weight'])
    # checkpoint round trip
    sd = m.state_dict()
    assert 'backbone.model.blocks.11.mlp.fc2.weight' in sd and 'kan_module.kan_layers.0.knots' in sd
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    m2 = RoViTKAN(pretrained=False, dropout=0.0).to(dev())
    m2.load_state_dict(torch.load(buf, weights_only=True))
    m.eval(); m2.eval()
    with torch.no_grad():
        o1, o2 = m(x), m2(x)
    for k in ('cls_logits', 'features', 'ordinal_logits', 'mu', 'log_var', 'kan_severity'):
        assert torch.equal(o1[k], o2[k]), k
