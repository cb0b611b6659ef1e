"""Round-4 GPU tests: the two-workgroups-per-CU attention kernels (csrc/attention.hip), ...

Arithmetic being checked: timm Attention.forward (softmax(q k^T / 8) v) and its autograd backward, reached through
/root/reference/models/backbone.py:23-25; the checker is a plain fp32 torch restatement on the same
bf16 inputs (floating-point kernel: tolerances stated per assertion)."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')
for p_ in (ROOT, PKG):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def bf(t):
    return t.to(torch.bfloat16)


def _native():
    from rovit_hip import native
    native.load()
    return native


def _attn_ref(qkv, B, Tk, H):
    q, k, v = qkv.float().view(B, Tk, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q * 0.125) @ k.transpose(-2, -1), dim=-1)
    return (a @ v).transpose(1, 2).reshape(B * Tk, H * 64)


@pytest.mark.parametrize('B,Tk', [(1, 197), (2, 197), (5, 197), (3, 64), (2, 208), (4, 33), (7, 1), (3, 16), (2, 17), (260, 197)])
def test_attention_two_workgroups_per_cu_ragged_sizes_vs_fp32_reference(B, Tk):
    """Forward and backward at ragged token counts (sub-tiles beyond T are skipped, the last key block is masked), batches below and
    above the chip's 512 workgroup slots; outputs poisoned with NaN first (every element must be written); bit-identical run to run."""
    native = _native()
    H = 3
    torch.manual_seed(B * 1000 + Tk)
    M = B * Tk
    qkv = bf(torch.randn(M, 3 * H * 64, device=dev()) * 1.3)
    dO = bf(torch.randn(M, H * 64, device=dev()))
    p, sp = native.ptr, native.stream_ptr()
    outs = []
    for rep in range(2):
        o = torch.full((M, H * 64), float('nan'), device=dev(), dtype=torch.bfloat16)
        lse = torch.full((B, H, Tk), float('nan'), device=dev())
        dqkv = torch.full((M, 3 * H * 64), float('nan'), device=dev(), dtype=torch.bfloat16)
        native.call('rovit_attention_fwd', p(qkv), p(o), p(lse), B, Tk, H, 64, 0.125, sp)
        native.call('rovit_attention_bwd', p(qkv), p(o), p(lse), p(dO), p(dqkv), B, Tk, H, 64, 0.125, sp)
        outs.append((o, lse, dqkv))
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2].view(torch.int16), outs[1][2].view(torch.int16))
    o, lse, dqkv = outs[0]
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all() and torch.isfinite(dqkv.float()).all()
    qf = qkv.float().requires_grad_(True)
    ref = _attn_ref(qf, B, Tk, H)
    assert float((o.float() - ref).abs().max()) < 2e-2                    # bf16 probabilities and output, |v| ~ 1.3
    q, k = qkv.float().view(B, Tk, 3, H, 64).permute(2, 0, 3, 1, 4)[:2]
    ref_lse = torch.logsumexp((q * 0.125) @ k.transpose(-2, -1), dim=-1) / math.log(2.0)
    assert float((lse - ref_lse).abs().max()) < 1e-3
    ref.backward(dO.float())
    assert float((dqkv.float() - qf.grad).abs().max()) < 3e-2 * float(qf.grad.abs().max())


def test_attention_backward_with_hugely_negative_logits_stays_finite():
    """with the key masks gone, a padded key (score 0) of a query row whose log-sum-exp is below about -125 made
    exp2(0 - lse) overflow, and Inf x 0 (the zero K row) = NaN landed in dQ.  Round 4 masks dS in the one key block that holds
    padded keys.  Rows 0..3 of q are anti-aligned with every key: scaled scores about -250 (lse2 about -360)."""
    native = _native()
    B, Tk, H = 2, 197, 3
    torch.manual_seed(11)
    M = B * Tk
    qkv = torch.randn(M, 3 * H * 64, device=dev()) * 0.5
    kdir = torch.ones(64, device=dev()) * 4.0
    qkv[:, 192:256] = kdir + 0.25 * torch.randn(M, 64, device=dev())       # every key of head 0 near +4 in every dim
    for r in range(4):
        qkv[r, 0:64] = -8.0                                                # q . k ~ -2048, x 0.125 = -256
    qkv = bf(qkv)
    dO = bf(torch.randn(M, H * 64, device=dev()))
    o = torch.empty(M, H * 64, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, Tk, device=dev())
    dqkv = torch.full((M, 3 * H * 64), float('nan'), device=dev(), dtype=torch.bfloat16)
    p, sp = native.ptr, native.stream_ptr()
    native.call('rovit_attention_fwd', p(qkv), p(o), p(lse), B, Tk, H, 64, 0.125, sp)
    assert float(lse[0, 0, :4].max()) < -300.0                             # the regime of the finding
    native.call('rovit_attention_bwd', p(qkv), p(o), p(lse), p(dO), p(dqkv), B, Tk, H, 64, 0.125, sp)
    assert torch.isfinite(dqkv.float()).all()
    qf = qkv.float().requires_grad_(True)
    _attn_ref(qf, B, Tk, H).backward(dO.float())
    assert float((dqkv.float() - qf.grad).abs().max()) < 3e-2 * float(qf.grad.abs().max())


# ------------------------------------------------------------------------------------------------------------------
# Round 4: the residual gradient travels between the backward's kernels as bf16 rows (csrc/vit.hip); the kernels sum in fp32.
# Arithmetic: autograd of timm Block (x = x + f(norm(x))): dX_out = dX_in + LayerNorm-backward(dgrad), reached through
# /root/reference/models/backbone.py:23-25 and training/trainer.py:119,136.
@pytest.mark.parametrize('M,K', [(394, 768), (1000, 576), (50, 192), (256 * 197, 576)])
def test_gemm_ln_bwd_with_a_bf16_residual_gradient_equals_the_fp32_form_on_the_same_rounded_input(M, K):
    """dXb_in given: dXb must be bit-identical to what the fp32 form produces from dX = float(dXb_in) (same arithmetic, same order:
    only where the incoming rows come from differs), and the fp32 dX must stay untouched."""
    native = _native()
    torch.manual_seed(M * 3 + K)
    dY = bf(torch.randn(M, K, device=dev()))
    W = bf(torch.randn(192, K, device=dev()) * 0.05)
    xh = bf(torch.randn(M, 192, device=dev()))
    rstd = torch.rand(M, device=dev()) + 0.5
    dXin = bf(torch.randn(M, 192, device=dev()))
    p, sp = native.ptr, native.stream_ptr()
    dX = dXin.float()
    ref_b = torch.empty(M, 192, device=dev(), dtype=torch.bfloat16)
    native.call('rovit_gemm_ln_bwd', p(dY), K, p(W), K, M, K, p(xh), p(rstd), p(dX), None, p(ref_b), sp)
    guard = torch.full((M, 192), 7.0, device=dev())
    out_b = torch.full((M, 192), float('nan'), device=dev(), dtype=torch.bfloat16)
    native.call('rovit_gemm_ln_bwd', p(dY), K, p(W), K, M, K, p(xh), p(rstd), p(guard), p(dXin), p(out_b), sp)
    assert torch.equal(out_b.view(torch.int16), ref_b.view(torch.int16))
    assert bool((guard == 7.0).all())
    out2 = torch.full((M, 192), float('nan'), device=dev(), dtype=torch.bfloat16)
    native.call('rovit_gemm_ln_bwd', p(dY), K, p(W), K, M, K, p(xh), p(rstd), None, p(dXin), p(out2), sp)     # dX may be NULL
    assert torch.equal(out2.view(torch.int16), ref_b.view(torch.int16))
    g = bf(dY.float() @ W.float().t()).float()
    h = xh.float()
    ref = dXin.float() + rstd[:, None] * (g - g.mean(1, keepdim=True) - h * (g * h).mean(1, keepdim=True))
    assert float((out_b.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())      # one bf16 rounding of the result


@pytest.mark.parametrize('M', [1, 257, 1000, 256 * 197])
def test_fused_mlp_backward_with_a_bf16_residual_gradient(M):
    """rovit_mlp_fused_bwd with dX = NULL: the incoming gradient is dY itself; dXb bit-identical to the fp32 form started from
    dX = float(dY), dpre unchanged."""
    native = _native()
    g = torch.Generator(device='cpu').manual_seed(77 + M)
    r = lambda *s: torch.randn(*s, generator=g)
    dY = bf(r(M, 192)).to(dev())
    w2t = bf(r(768, 192) * 0.05).to(dev())
    w1t = bf(r(192, 768) * 0.05).to(dev())
    dact = bf(torch.rand(M, 768, generator=g) * 1.2 - 0.1).to(dev())
    dact_c = dact.view(M, 24, 32).permute(1, 0, 2).contiguous().view(M, 768)
    xh = bf(r(M, 192)).to(dev())
    rstd = (torch.rand(M, generator=g) + 0.5).to(dev())
    p, sp = native.ptr, native.stream_ptr()
    ws = torch.empty(native.load().rovit_mlp_stream_bytes(), dtype=torch.uint8, device=dev())
    native.call('rovit_mlp_prepare_stream', p(w2t), p(w1t), p(ws), sp)
    outs = []
    for mode in ('fp32', 'bf16'):
        dpre = torch.full((M, 768), float('nan'), device=dev(), dtype=torch.bfloat16)
        dXb = torch.full((M, 192), float('nan'), device=dev(), dtype=torch.bfloat16)
        dX = dY.float() if mode == 'fp32' else None
        native.call('rovit_mlp_fused_bwd', p(dY), p(ws), p(dact_c), p(dpre), p(xh), p(rstd), p(dX), p(dXb), M, sp)
        outs.append((dpre, dXb))
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))
    assert torch.isfinite(outs[1][1].float()).all()


def test_pos_grad_from_bf16_rows_and_cls_norm_bwd_without_zero_fill():
    native = _native()
    B, T = 5, 197
    torch.manual_seed(3)
    p, sp = native.ptr, native.stream_ptr()
    dXb = bf(torch.randn(B * T, 192, device=dev()))
    dpos = torch.empty(T, 192, device=dev())
    dcls = torch.empty(192, device=dev())
    native.call('rovit_pos_grad', None, p(dXb), p(dpos), p(dcls), B, T, sp)
    ref = dXb.float().view(B, T, 192).sum(0)
    assert float((dpos - ref).abs().max()) < 1e-5 and torch.equal(dcls, dpos[0])
    dpos2 = torch.empty_like(dpos)
    native.call('rovit_pos_grad', p(dXb.float()), None, p(dpos2), p(dcls), B, T, sp)
    assert torch.equal(dpos, dpos2)
    with pytest.raises(native.RovitHipError):
        native.call('rovit_pos_grad', None, None, p(dpos), p(dcls), B, T, sp)
    # final-norm backward: zero_fill = 0 leaves the other rows alone, zero_fill = 1 zeroes them; CLS rows identical
    dfeat = torch.randn(B, 192, device=dev())
    xhat = torch.randn(B, 192, device=dev())
    rstd = torch.rand(B, device=dev()) + 0.5
    gamma = torch.randn(192, device=dev())
    res = []
    for zf in (1, 0):
        dX = torch.full((B * T, 192), 3.0, device=dev())
        dXo = torch.full((B * T, 192), 3.0, device=dev(), dtype=torch.bfloat16)
        dg, db = torch.empty(192, device=dev()), torch.empty(192, device=dev())
        native.call('rovit_cls_norm_bwd', p(dfeat), p(xhat), p(rstd), p(gamma), p(dX), p(dXo), p(dg), p(db), B, T, zf, sp)
        res.append((dX.view(B, T, 192), dXo.view(B, T, 192)))
    assert torch.equal(res[0][0][:, 0], res[1][0][:, 0]) and torch.equal(res[0][1][:, 0], res[1][1][:, 0])
    assert bool((res[0][0][:, 1:] == 0).all()) and bool((res[1][0][:, 1:] == 3.0).all()) and bool((res[1][1][:, 1:].float() == 3.0).all())


@pytest.mark.parametrize('B,T', [(5, 197), (3, 64), (2, 208), (1, 9)])
def test_class_token_attention_equals_the_full_kernels_row_and_a_torch_reference(B, T):
    """The last block's attention (rovit_attention_cls_fwd / _bwd): only the class token's output is consumed, so only its query is
    evaluated.  Against fp32 torch on the same bf16 operands: 2e-2 of the bf16 output scale forward (the full kernel's own distance),
    and backward gradients within 2e-2 relative; against the full kernels' class-token row / a dout that is zero elsewhere: the same
    bound (the full kernel rounds the probabilities to bf16 for its matrix products, this one keeps them in fp32).  Every dQ row but
    the class token's must be EXACT zeros (the buffer is reused across blocks: poisoned first), whatever tokens."""
    from rovit_hip import native
    H, HD = 3, 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * H * HD, generator=g) * 1.5).to(dev()).to(torch.bfloat16)
    dout = torch.zeros(B * T, H * HD, device=dev(), dtype=torch.bfloat16)
    dcls = torch.randn(B, H * HD, generator=g).to(dev()).to(torch.bfloat16)
    dout.view(B, T, H * HD)[:, 0] = dcls
    sp = native.stream_ptr()
    # fp32 reference on the bf16 values
    x = qkv.float().view(B, T, 3, H, HD)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(3))        # (B,H,T,HD)
    s = (q[:, :, :1] @ k.transpose(-1, -2)) * 0.125
    o_ref = torch.softmax(s, dim=-1) @ v                                                                # (B,H,1,HD)
    (o_ref * dcls.float().view(B, 1, H, HD).permute(0, 2, 1, 3)).sum().backward()
    # full kernels
    out_full = torch.empty(B * T, H * HD, device=dev(), dtype=torch.bfloat16)
    lse_full = torch.empty(B, H, T, device=dev())
    native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out_full), native.ptr(lse_full), B, T, H, HD, 0.125, sp)
    dq_full = torch.empty_like(qkv)
    native.call('rovit_attention_bwd', native.ptr(qkv), native.ptr(out_full), native.ptr(lse_full), native.ptr(dout), native.ptr(dq_full),
                B, T, H, HD, 0.125, sp)
    # class-token kernels (outputs poisoned first)
    out = torch.full((B * T, H * HD), float('nan'), device=dev(), dtype=torch.bfloat16)
    lse = torch.full((B, H, T), float('nan'), device=dev())
    native.call('rovit_attention_cls_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, H, HD, 0.125, sp)
    o = out.view(B, T, H, HD)[:, 0].float()
    ref = o_ref[:, :, 0].detach().float()
    scale = float(ref.abs().max())
    assert float((o - ref).abs().max()) < 1e-2 * scale + 1e-2
    assert float((o - out_full.view(B, T, H, HD)[:, 0].float()).abs().max()) < 2e-2 * scale + 1e-2
    assert float((lse[:, :, 0] - lse_full[:, :, 0]).abs().max()) < 2e-2
    assert bool(torch.isnan(out.view(B, T, H * HD)[:, 1:].float()).all()) or T == 1        # nothing but the class token's row is written
    dqkv = torch.full_like(qkv, float('nan'))
    native.call('rovit_attention_cls_bwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), native.ptr(dout), native.ptr(dqkv), B, T, H, HD,
                0.125, sp)
    d = dqkv.float().view(B, T, 3, H, HD)
    assert not bool(torch.isnan(d).any())
    assert bool((d[:, 1:, 0] == 0).all())                                  # exact zeros in every other query's gradient
    for i, ref_g in enumerate((q.grad, k.grad, v.grad)):
        got = d[:, :, i].permute(0, 2, 1, 3)
        sc = float(ref_g.abs().max())
        assert float((got - ref_g).abs().max()) < 2e-2 * sc + 1e-3, 'qkv'[i]
        full = dq_full.float().view(B, T, 3, H, HD)[:, :, i].permute(0, 2, 1, 3)
        assert float((got - full).abs().max()) < 3e-2 * sc + 1e-3, 'qkv'[i]
