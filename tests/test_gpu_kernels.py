"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes), against plain PyTorch
fp32 references of the same op and against the golden vectors generated from the reference."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu  # noqa: E402  (checker only)

T = torch.from_numpy


def _native():
    from rovit_hip import native
    native.load()
    return native


def dev():
    return torch.device('cuda:0')


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def bf(x):
    return x.to(torch.bfloat16)


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6))


# ------------------------------------------------------------------ KAN ------------------------------------
def test_kan_basis_table(golden_dir):
    from models.kan import BSplineBasis
    g = load(golden_dir, 'kan_basis')
    for G in (5, 32):
        x, knots, ref = T(g[f'g{G}.x']).to(dev()), T(g[f'g{G}.knots']).to(dev()), T(g[f'g{G}.basis'])
        got = BSplineBasis.compute_basis(x, knots, 3).cpu()
        assert got.shape == ref.shape
        # off-knot points: tight; the truncation to zero beyond knots[num_basis] must be exact
        assert float((got - ref).abs().max()) < 2e-6
        beyond = T(g[f'g{G}.x']) >= T(g[f'g{G}.knots'])[knots.numel() - 4]
        assert float(got[beyond].abs().max()) == 0.0


@pytest.mark.parametrize('name', ['kan_mini', 'kan_default', 'kan_g32'])
def test_kan_module_vs_reference_golden(golden_dir, name):
    from models.kan import KANSeverityModule
    g = load(golden_dir, name)
    layers = [int(v) for v in g['layers']]
    m = KANSeverityModule(layers, int(g['num_knots']), int(g['degree']))
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith('sd.')})
    m = m.to(dev())
    x = T(g['x']).to(dev()).requires_grad_(True)
    y = m(x)
    assert float((y.cpu() - T(g['y'])).abs().max()) < 1e-4        # north_star: severity within 1e-3
    traj = m.get_activation_trajectory(x.detach())
    for i, t in enumerate(traj):
        assert float((t.cpu() - T(g[f'traj.{i}'])).abs().max()) < 1e-4, i
    (y * T(g['w']).to(dev())).sum().backward()
    ref_dx = T(g['dx'])
    assert float((x.grad.cpu() - ref_dx).abs().max()) < 1e-4 * max(1.0, float(ref_dx.abs().max()))
    for k, p in m.named_parameters():
        ref = T(g['grad.' + k])
        assert float((p.grad.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), k


def test_kan_visualisation_helpers_match_the_reference_formulas():
    """KANLayer.plot_activation / get_spline_weights and KANSeverityModule.get_spline_weights (models/kan.py:96-114,151), what
    explainability/kan_viz.py:22-36,184 draws: y(x) = sum_k basis_k(x) W[in, out, k] on linspace(-1, 1, n) WITHOUT tanh, zero
    from the cutoff knot on."""
    from models.kan import KANSeverityModule
    g = torch.Generator().manual_seed(9)
    sd = ref_cpu.init_kan_state([16, 8, 1], 5, 3, g)
    m = KANSeverityModule([16, 8, 1], 5, 3)
    m.load_state_dict(sd)
    m = m.to(dev())
    layer = m.kan_layers[0]
    for i, o, n in ((0, 0, 100), (3, 5, 41), (15, 7, 7)):
        xs, ys = layer.plot_activation(i, o, n)
        xr = torch.linspace(-1, 1, n)
        basis = ref_cpu.truncated_bspline_basis(xr.unsqueeze(0), sd['kan_layers.0.knots'], 3)[0]
        yr = (basis * sd['kan_layers.0.spline_weights'][i, o]).sum(dim=1)
        assert xs.shape == (n,) and np.allclose(xs, xr.numpy()) and np.abs(ys - yr.numpy()).max() < 1e-6
        assert np.all(ys[xs >= ref_cpu_cut(sd)] == 0.0)
    w = layer.get_spline_weights()
    assert not w.requires_grad and torch.equal(w.cpu(), sd['kan_layers.0.spline_weights'])
    assert [tuple(t.shape) for t in m.get_spline_weights()] == [(16, 8, 7), (8, 1, 7)]


def ref_cpu_cut(sd):
    k = sd['kan_layers.0.knots']
    return float(k[k.numel() - 4])          # knots[num_basis]: the basis is identically zero from here on


def test_kan_large_batch_vs_oracle():
    """BASELINE config 5 shape: grid_size=32, 3 stacked layers, batch 512."""
    from models.kan import KANSeverityModule
    torch.manual_seed(5)
    m = KANSeverityModule([192, 64, 16, 1], 32, 3)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(512, 192)
    ref = ref_cpu.kan_module_forward(x, sd, degree=3)
    y = m.to(dev())(x.to(dev()))
    assert float((y.cpu() - ref).abs().max()) < 1e-4


def test_kan_saturated_inputs():
    from models.kan import KANLayer
    torch.manual_seed(1)
    layer = KANLayer(8, 4, 5, 3)
    x = torch.tensor([[-50.0, -9.0, -0.4236, 0.0, 0.4236, 0.5, 9.0, 50.0]] * 3)
    sd = {'kan_layers.0.' + k: v for k, v in layer.state_dict().items()}
    ref = ref_cpu.kan_layer_forward(x, sd['kan_layers.0.spline_weights'], sd['kan_layers.0.knots'],
                                    sd['kan_layers.0.linear.weight'], sd['kan_layers.0.linear.bias'])
    y = layer.to(dev())(x.to(dev()))
    assert torch.isfinite(y).all()
    assert float((y.cpu() - ref).abs().max()) < 1e-4


# ------------------------------------------------------------------ heads ----------------------------------
def test_heads_vs_reference_golden(golden_dir):
    from models.heads import ClassificationHead, OrdinalHead, UncertaintyHead
    g = load(golden_dir, 'heads')
    mods = {'classification_head': ClassificationHead(192, 128, 4, 0.3), 'ordinal_head': OrdinalHead(192, 128, 4, 0.3),
            'uncertainty_head': UncertaintyHead(192, 128, 0.3)}
    for name, m in mods.items():
        m.load_state_dict({k[len('sd.' + name) + 1:]: T(g[k]) for k in g.files if k.startswith('sd.' + name + '.')})
        m.to(dev()).eval()
    x = T(g['x']).to(dev()).requires_grad_(True)
    cl, ol = mods['classification_head'](x), mods['ordinal_head'](x)
    mu, lv = mods['uncertainty_head'](x)
    for got, key in ((cl, 'cls_logits'), (ol, 'ordinal_logits'), (mu, 'mu'), (lv, 'log_var')):
        assert float((got.cpu() - T(g[key])).abs().max()) < 1e-4, key
    assert float((mods['ordinal_head'].predict_severity(x).cpu() - T(g['ord_severity'])).abs().max()) < 1e-4
    loss = sum((o * T(g[w]).to(dev())).sum() for o, w in ((cl, 'w.cls'), (ol, 'w.ord'), (mu, 'w.mu'), (lv, 'w.lv')))
    loss.backward()
    assert float((x.grad.cpu() - T(g['dx'])).abs().max()) < 1e-4
    for name, m in mods.items():
        for k, p in m.named_parameters():
            assert float((p.grad.cpu() - T(g[f'grad.{name}.{k}'])).abs().max()) < 2e-4, (name, k)


def test_heads_dropout_mask_and_clamp():
    """train-mode semantics with an injected mask; log_var clamp gradient gate."""
    native = _native()
    from rovit_hip.functions import MLPHeadFn
    torch.manual_seed(0)
    B = 16
    x = torch.randn(B, 192)
    w1, b1 = torch.randn(128, 192) * 0.1, torch.randn(128) * 0.1
    w2, b2 = torch.randn(1, 128) * 3.0, torch.randn(1)
    mask = (torch.rand(B, 128) < 0.7).float() / 0.7
    xs = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    h = torch.relu(torch.nn.functional.linear(xs[0], xs[1], xs[2])) * mask
    ref = torch.clamp(torch.nn.functional.linear(h, xs[3], xs[4]), -10, 10)
    assert (ref.abs() == 10).any() and (ref.abs() < 10).any()
    ref.sum().backward()
    ys = [t.clone().to(dev()).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    (got,) = MLPHeadFn.apply(ys[0], mask.to(dev()), (2,), ys[1], ys[2], ys[3], ys[4])
    assert float((got.cpu() - ref).abs().max()) < 1e-4
    got.sum().backward()
    for a, b in zip(ys, xs):
        assert float((a.grad.cpu() - b.grad).abs().max()) < 2e-4 * max(1.0, float(b.grad.abs().max()))


# ------------------------------------------------------------------ GEMM -----------------------------------
def _gemm(native, A, W, bias, epi, **kw):
    M, K = A.shape
    N = W.shape[0]
    out = kw.get('out')
    out2 = kw.get('out2')
    xres = kw.get('xres')
    mul = kw.get('mul')
    pos = kw.get('pos')
    native.call('rovit_gemm_nt', native.ptr(A), A.stride(0), native.ptr(W), W.stride(0), M, N, K, native.ptr(bias), epi,
                native.ptr(out), out.stride(0) if out is not None else 0, native.ptr(out2), native.ptr(xres),
                xres.stride(0) if xres is not None else 0, native.ptr(mul), mul.stride(0) if mul is not None else 0,
                native.ptr(pos), kw.get('tokens', 0), native.stream_ptr())


@pytest.mark.parametrize('tile', [0, 0x100, 0x200])      # weight-stationary kernels; ROVIT_GEMM_TILED_192 / _96: the LDS-tiled kernels, per call
@pytest.mark.parametrize('M,N,K', [(394, 576, 192), (1000, 192, 768), (128, 768, 192), (77, 192, 576), (2561, 192, 192)])
def test_gemm_bf16_bias(M, N, K, tile):
    native = _native()
    torch.manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device=dev()))
    W = bf(torch.randn(N, K, device=dev()) * 0.05)
    bias = torch.randn(N, device=dev())
    # asymmetric integer-valued check first: exact in bf16/fp32, catches any fragment-layout transposition
    Ai = bf(torch.randint(-3, 4, (M, K), device=dev()).float())
    Wi = bf(torch.randint(-2, 3, (N, K), device=dev()).float())
    out = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    _gemm(native, Ai, Wi, None, 0 | tile, out=out)
    ref = Ai.float() @ Wi.float().t()
    assert torch.equal(out.float(), bf(ref).float())
    _gemm(native, A, W, bias, 0 | tile, out=out)
    ref = A.float() @ W.float().t() + bias
    assert relerr(out, ref) < 1e-2


def test_gemm_epilogues():
    native = _native()
    torch.manual_seed(0)
    M, N, K = 394, 768, 192
    A = bf(torch.randn(M, K, device=dev()))
    W = bf(torch.randn(N, K, device=dev()) * 0.08)
    bias = torch.randn(N, device=dev()) * 0.5
    pre = A.float() @ W.float().t() + bias
    # GELU: act and its derivative
    act = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    dact = torch.empty_like(act)
    _gemm(native, A, W, bias, 1, out=act, out2=dact)
    p = pre.clone().requires_grad_(True)
    ref_act = torch.nn.functional.gelu(p)
    ref_act.sum().backward()
    assert float((act.float() - ref_act).abs().max()) < 3e-2
    assert float((dact.float() - p.grad).abs().max()) < 1e-2
    # residual (fp32, in place)
    N2 = 192
    W2 = bf(torch.randn(N2, K, device=dev()) * 0.08)
    b2 = torch.randn(N2, device=dev())
    X = torch.randn(M, N2, device=dev())
    ref = X + A.float() @ W2.float().t() + b2
    _gemm(native, A, W2, b2, 2, xres=X)
    assert float((X - ref).abs().max()) < 2e-3
    # multiply
    mul = bf(torch.rand(M, N, device=dev()))
    out = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    _gemm(native, A, W, None, 3, out=out, mul=mul)
    ref = (A.float() @ W.float().t()) * mul.float()
    assert relerr(out, ref) < 1e-2
    # patch scatter: rows (b, p) -> token rows b*T+1+p, + pos
    B, Tk = 2, 197
    col = bf(torch.randn(B * 196, 768, device=dev()))
    Wp = bf(torch.randn(192, 768, device=dev()) * 0.03)
    bp = torch.randn(192, device=dev())
    pos = torch.randn(Tk, 192, device=dev())
    Xt = torch.full((B * Tk, 192), 7.0, device=dev())
    _gemm(native, col, Wp, bp, 4, xres=Xt, pos=pos, tokens=Tk)
    ref = (col.float() @ Wp.float().t() + bp).view(B, 196, 192) + pos[1:]
    got = Xt.view(B, Tk, 192)
    assert float((got[:, 1:] - ref).abs().max()) < 2e-3
    assert float((got[:, 0] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize('M,N,K', [(394, 576, 192), (1970, 192, 768), (50, 768, 192), (3333, 192, 192)])
def test_wgrad(M, N, K):
    native = _native()
    lib = native.load()
    torch.manual_seed(M)
    dY = bf(torch.randn(M, N, device=dev()))
    A = bf(torch.randn(M, K, device=dev()))
    splits = lib.rovit_wgrad_splits(M, N, K)
    ws = torch.empty(lib.rovit_wgrad_workspace_bytes(N, K, splits) // 4, device=dev())
    dW, db = torch.empty(N, K, device=dev()), torch.empty(N, device=dev())

    def run(dy, a):
        native.call('rovit_wgrad', native.ptr(dy), dy.stride(0), native.ptr(a), a.stride(0), M, N, K, splits, 0, native.ptr(ws),
                    native.stream_ptr())
        native.call('rovit_wgrad_reduce', native.ptr(ws), splits, N, K, None, None, None, native.ptr(dW), native.ptr(db), None,
                    None, None, native.stream_ptr())
    Yi = bf(torch.randint(-2, 3, (M, N), device=dev()).float())
    Ai = bf(torch.randint(-3, 4, (M, K), device=dev()).float())
    run(Yi, Ai)
    assert torch.equal(dW, Yi.float().t() @ Ai.float())             # exact integer check (asymmetric operands)
    assert torch.equal(db, Yi.float().sum(0))
    run(dY, A)
    ref = dY.float().t() @ A.float()
    assert float((dW - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    assert float((db - dY.float().sum(0)).abs().max()) < 1e-2
    # folded-affine un-fold: W_f = W*gamma, b_f = b + W beta
    gamma, beta = torch.randn(K, device=dev()), torch.randn(K, device=dev())
    Wt = torch.randn(N, K, device=dev())
    dgam, dbet, gs = torch.empty(K, device=dev()), torch.empty(K, device=dev()), torch.empty(N, K, device=dev())
    native.call('rovit_wgrad_reduce', native.ptr(ws), splits, N, K, native.ptr(gamma), native.ptr(beta), native.ptr(Wt),
                native.ptr(dW), native.ptr(db), native.ptr(dgam), native.ptr(dbet), native.ptr(gs), native.stream_ptr())
    G, cb = ref, dY.float().sum(0)
    tol = 3e-3 * float(G.abs().max()) * float(gamma.abs().max())
    assert float((dW - (G * gamma + cb[:, None] * beta)).abs().max()) < tol
    assert float((dgam - (Wt * G).sum(0)).abs().max()) < 3e-3 * float((Wt * G).sum(0).abs().max())
    assert float((dbet - (Wt * cb[:, None]).sum(0)).abs().max()) < 3e-3 * float((Wt * cb[:, None]).sum(0).abs().max())


def test_wgrad_patch_rows():
    native = _native()
    lib = native.load()
    B, Tk = 3, 197
    dX = bf(torch.randint(-2, 3, (B * Tk, 192), device=dev()).float())
    col = bf(torch.randint(-2, 3, (B * 196, 768), device=dev()).float())
    M = B * 196
    splits = lib.rovit_wgrad_splits(M, 192, 768)
    ws = torch.empty(lib.rovit_wgrad_workspace_bytes(192, 768, splits) // 4, device=dev())
    dW, db = torch.empty(192, 768, device=dev()), torch.empty(192, device=dev())
    native.call('rovit_wgrad', native.ptr(dX), 192, native.ptr(col), 768, M, 192, 768, splits, Tk, native.ptr(ws), native.stream_ptr())
    native.call('rovit_wgrad_reduce', native.ptr(ws), splits, 192, 768, None, None, None, native.ptr(dW), native.ptr(db), None, None,
                None, native.stream_ptr())
    dy = dX.view(B, Tk, 192)[:, 1:].reshape(M, 192).float()
    assert torch.equal(dW, dy.t() @ col.float())
    assert torch.equal(db, dy.sum(0))


# ------------------------------------------------------------------ LayerNorm / im2col ---------------------
def test_layernorm_fwd_bwd():
    native = _native()
    torch.manual_seed(0)
    M = 1000
    x = (torch.randn(M, 192, device=dev()) * 3 + 1).requires_grad_(True)
    xhat = torch.empty(M, 192, device=dev(), dtype=torch.bfloat16)
    rstd = torch.empty(M, device=dev())
    native.call('rovit_layernorm_fwd', native.ptr(x.detach()), native.ptr(xhat), native.ptr(rstd), M, 192, 1e-6, native.stream_ptr())
    ref = torch.nn.functional.layer_norm(x, (192,), eps=1e-6)
    assert float((xhat.float() - ref).abs().max()) < 2e-2
    assert relerr(rstd, 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)) < 1e-5
    g = bf(torch.randn(M, 192, device=dev()))
    dX0 = torch.randn(M, 192, device=dev())
    # reference backward evaluated at the bf16 xhat the kernel consumes
    xh = xhat.float()
    gf = g.float()
    ref_dx = dX0 + rstd[:, None] * (gf - gf.mean(1, keepdim=True) - xh * (gf * xh).mean(1, keepdim=True))
    dX = dX0.clone()
    dXb = torch.empty(M, 192, device=dev(), dtype=torch.bfloat16)
    native.call('rovit_layernorm_bwd', native.ptr(g), native.ptr(xhat), native.ptr(rstd), native.ptr(dX), native.ptr(dXb), M, 192,
                native.stream_ptr())
    assert float((dX - ref_dx).abs().max()) < 1e-4 * float(ref_dx.abs().max())
    assert torch.equal(dXb, bf(dX))
    # and against autograd of the fp32 op (tolerance = bf16 rounding of xhat)
    ref.backward(gf)
    assert float((dX - dX0 - x.grad).abs().max()) < 3e-2 * float(x.grad.abs().max())


def test_im2col_matches_conv():
    native = _native()
    torch.manual_seed(0)
    B = 3
    x = torch.randn(B, 3, 224, 224, device=dev())
    col = torch.empty(B * 196, 768, device=dev(), dtype=torch.bfloat16)
    native.call('rovit_im2col', native.ptr(x), native.ptr(col), B, native.stream_ptr())
    ref = torch.nn.functional.unfold(x, kernel_size=16, stride=16).transpose(1, 2).reshape(B * 196, 768)
    assert torch.equal(col, bf(ref))


@pytest.mark.parametrize('B', [1, 3, 37])
def test_patch_embed_without_im2col_buffer_is_bit_identical_to_the_im2col_path(B):
    """rovit_patch_embed_fwd / rovit_patch_embed_wgrad gather the conv patches from the fp32 images inside the GEMM and the
    weight-gradient kernel; they must give the bits of rovit_im2col + rovit_gemm_nt(EPI_PATCH) / rovit_wgrad(patch rows), and
    the values of the conv itself (PatchEmbed of timm's VisionTransformer, reference models/backbone.py:12-25)."""
    native = _native()
    lib = native.load()
    g = torch.Generator(device=dev()).manual_seed(B)
    Tk = 197
    x = torch.randn(B, 3, 224, 224, device=dev(), generator=g)
    W = bf(torch.randn(192, 768, device=dev(), generator=g) * 0.05)
    bias = torch.randn(192, device=dev(), generator=g)
    pos = torch.randn(Tk, 192, device=dev(), generator=g)
    col = torch.empty(B * 196, 768, device=dev(), dtype=torch.bfloat16)
    native.call('rovit_im2col', native.ptr(x), native.ptr(col), B, native.stream_ptr())
    X0 = torch.full((B * Tk, 192), 7.0, device=dev())
    X1 = X0.clone()
    native.call('rovit_gemm_nt', native.ptr(col), 768, native.ptr(W), 768, B * 196, 192, 768, native.ptr(bias), 4, None, 0, None,
                native.ptr(X0), 192, None, 0, native.ptr(pos), Tk, native.stream_ptr())
    native.call('rovit_patch_embed_fwd', native.ptr(x), native.ptr(W), native.ptr(bias), native.ptr(pos), native.ptr(X1), B, Tk,
                native.stream_ptr())
    assert torch.equal(X0, X1)
    conv = torch.nn.functional.conv2d(bf(x).float(), W.float().view(192, 3, 16, 16), bias, stride=16).flatten(2).transpose(1, 2)
    got = X1.view(B, Tk, 192)[:, 1:] - pos[1:]
    assert float((got - conv).abs().max()) < 2e-3 * float(conv.abs().max())
    assert torch.equal(X1.view(B, Tk, 192)[:, 0], torch.full((B, 192), 7.0, device=dev()))       # cls rows untouched
    # weight gradient
    dX = bf(torch.randn(B * Tk, 192, device=dev(), generator=g))
    M = B * 196
    splits = lib.rovit_wgrad_splits(M, 192, 768)
    nws = lib.rovit_wgrad_workspace_bytes(192, 768, splits) // 4
    ws0, ws1 = torch.empty(nws, device=dev()), torch.empty(nws, device=dev())
    native.call('rovit_wgrad', native.ptr(dX), 192, native.ptr(col), 768, M, 192, 768, splits, Tk, native.ptr(ws0), native.stream_ptr())
    native.call('rovit_patch_embed_wgrad', native.ptr(dX), 192, native.ptr(x), B, Tk, 192, splits, native.ptr(ws1), native.stream_ptr())
    assert torch.equal(ws0, ws1)


# ------------------------------------------------------------------ attention -------------------------------
def _attn_ref(qkv, B, Tk, H):
    q, k, v = qkv.float().view(B, Tk, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q * 0.125) @ k.transpose(-2, -1), dim=-1)
    return (a @ v).transpose(1, 2).reshape(B * Tk, H * 64), a


@pytest.mark.parametrize('B,Tk', [(2, 197), (3, 50), (1, 208), (5, 197)])
def test_attention_fwd_bwd(B, Tk):
    native = _native()
    torch.manual_seed(B * 1000 + Tk)
    H = 3
    qkv = bf(torch.randn(B * Tk, 3 * H * 64, device=dev()) * 1.5)
    out = torch.empty(B * Tk, H * 64, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, Tk, device=dev())
    native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, Tk, H, 64, 0.125, native.stream_ptr())
    qf = qkv.float().requires_grad_(True)
    ref, a = _attn_ref(qf, B, Tk, H)
    assert float((out.float() - ref).abs().max()) < 2e-2
    q, k = qkv.float().view(B, Tk, 3, H, 64).permute(2, 0, 3, 1, 4)[:2]
    ref_lse = torch.logsumexp((q * 0.125) @ k.transpose(-2, -1), dim=-1) / math.log(2.0)
    assert float((lse - ref_lse).abs().max()) < 1e-3
    dout = bf(torch.randn(B * Tk, H * 64, device=dev()))
    dqkv = torch.empty_like(qkv)
    native.call('rovit_attention_bwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), native.ptr(dout), native.ptr(dqkv), B, Tk, H,
                64, 0.125, native.stream_ptr())
    ref.backward(dout.float())
    assert float((dqkv.float() - qf.grad).abs().max()) < 3e-2 * float(qf.grad.abs().max())


def test_attention_spiked_scores():
    """one key dominating a row (large logits): softmax must stay finite and exact to bf16."""
    native = _native()
    B, Tk, H = 1, 197, 3
    torch.manual_seed(3)
    qkv = torch.randn(B * Tk, 576, device=dev())
    qkv[5, 0:64] = 6.0
    qkv[100, 192:256] = 6.0                     # q_5 . k_100 = 64*36*0.125 = 288
    qkv = bf(qkv)
    out = torch.empty(B * Tk, 192, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, Tk, device=dev())
    native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, Tk, H, 64, 0.125, native.stream_ptr())
    ref, _ = _attn_ref(qkv, B, Tk, H)
    assert torch.isfinite(out.float()).all()
    assert float((out.float() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize('M,K', [(394, 192), (1000, 768), (77, 576)])
def test_gemm_resid_ln_fused(M, K):
    """X += bf16(A W^T + b) fused with the following LayerNorm (timm Block: x = x + f(x); norm(x))."""
    native = _native()
    torch.manual_seed(M + K)
    A = bf(torch.randn(M, K, device=dev()))
    W = bf(torch.randn(192, K, device=dev()) * 0.05)
    bias = torch.randn(192, device=dev())
    X0 = torch.randn(M, 192, device=dev()) * 2
    X = X0.clone()
    xhat = torch.empty(M, 192, device=dev(), dtype=torch.bfloat16)
    rstd = torch.empty(M, device=dev())
    native.call('rovit_gemm_resid_ln', native.ptr(A), K, native.ptr(W), K, M, K, native.ptr(bias), native.ptr(X), native.ptr(xhat),
                native.ptr(rstd), 1e-6, native.stream_ptr())
    ref_x = X0 + bf(A.float() @ W.float().t() + bias).float()
    assert float((X - ref_x).abs().max()) < 2e-2                      # one bf16 ulp of the branch output
    ref_h = torch.nn.functional.layer_norm(X, (192,), eps=1e-6)       # statistics of the row the kernel wrote
    assert float((xhat.float() - ref_h).abs().max()) < 2e-2
    assert relerr(rstd, 1 / torch.sqrt(X.var(1, unbiased=False) + 1e-6)) < 1e-5
    X2 = X0.clone()                                                   # no LayerNorm requested: plain residual add
    native.call('rovit_gemm_resid_ln', native.ptr(A), K, native.ptr(W), K, M, K, native.ptr(bias), native.ptr(X2), None, None, 1e-6,
                native.stream_ptr())
    # (for K = 576 the two calls run different kernels -- K split over wave pairs vs. one wave over the whole K -- so the
    # fp32 summation order differs and a branch output may round to the neighbouring bf16)
    assert float((X2 - X).abs().max()) < 2e-2 and float((X2 - X).abs().mean()) < 1e-4


@pytest.mark.parametrize('M,K', [(394, 768), (1000, 576), (50, 192)])
def test_gemm_ln_bwd_fused(M, K):
    native = _native()
    torch.manual_seed(M * 3 + K)
    dY = bf(torch.randn(M, K, device=dev()))
    W = bf(torch.randn(192, K, device=dev()) * 0.05)
    xh = bf(torch.randn(M, 192, device=dev()))
    rstd = torch.rand(M, device=dev()) + 0.5
    dX0 = torch.randn(M, 192, device=dev())
    dX = dX0.clone()
    dXb = torch.empty(M, 192, device=dev(), dtype=torch.bfloat16)
    native.call('rovit_gemm_ln_bwd', native.ptr(dY), K, native.ptr(W), K, M, K, native.ptr(xh), native.ptr(rstd), native.ptr(dX),
                None, native.ptr(dXb), native.stream_ptr())
    g = bf(dY.float() @ W.float().t()).float()                        # the dgrad output, rounded as the kernel stages it
    h = xh.float()
    ref = dX0 + rstd[:, None] * (g - g.mean(1, keepdim=True) - h * (g * h).mean(1, keepdim=True))
    assert float((dX - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    assert torch.equal(dXb, bf(dX))
