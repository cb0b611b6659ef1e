"""KAN head timing (device events on the launch stream): the fused stack forward (rovit_kan_stack_fwd), the per-layer
forward kernels it replaces, and the backward, at the shapes BASELINE.json names.  Prints one JSON line per shape.

    python tools/bench_kan.py            # C3 head (G=5, B=256), C5 (G=32, B=512), streaming shapes B=65536
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch  # noqa: E402

from models.kan import KANSeverityModule  # noqa: E402
from rovit_hip.functions import ACT_RELU, ACT_SIGMOID3  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device('cuda:0')


def kan_bytes(layers, nb, B, fused=True):
    """algorithmic bytes of one forward: weights once per launch + activations"""
    w = sum(a * b * nb + a * b + b for a, b in zip(layers[:-1], layers[1:])) * 4
    if fused:
        act = B * (layers[0] + sum(layers[1:])) * 4                    # read x, write every layer output once
    else:
        act = B * (layers[0] + 2 * sum(layers[1:-1]) + layers[-1]) * 4
    return w + act


def main():
    torch.manual_seed(0)
    shapes = (('C3 head', [192, 64, 16, 1], 5, 256), ('C5 kan-heavy', [192, 64, 16, 1], 32, 512),
              ('streaming G=5', [192, 64, 16, 1], 5, 65536), ('streaming G=32', [192, 64, 16, 1], 32, 65536))
    if '--sweep' in sys.argv:          # where the three forward kernels cross over
        shapes = tuple((f'G={G} B={B}', [192, 64, 16, 1], G, B) for G in (5, 32) for B in (1024, 2048, 4096, 8192, 16384, 32768))
    for name, layers, G, B in shapes:
        m = KANSeverityModule(layers, G, 3).to(dev)
        nb = G + 2
        x = torch.randn(B, layers[0], device=dev)
        # kernels only: direct C-ABI calls with preallocated outputs (a module call costs ~100 us of Python per forward,
        # more than the kernels at batch 256)
        import ctypes as C
        from rovit_hip import native
        from rovit_hip.native import ptr, ptr_array
        prep = m._prepared()
        n = len(m.kan_layers)
        outs = [torch.empty(B, layers[l + 1], device=dev) for l in range(n)]
        arr = lambda xs: (C.c_int * len(xs))(*xs)
        lib = native.load()
        a_w, a_k, a_lw, a_lb, a_o = (ptr_array([p[0] for p in prep]), ptr_array([l.knots for l in m.kan_layers]), ptr_array([p[1] for p in prep]),
                                     ptr_array([l.linear.bias for l in m.kan_layers]), ptr_array(outs))
        dims, nks, acts = arr(layers), arr([l.knots.numel() for l in m.kan_layers]), arr([ACT_RELU] * (n - 1) + [ACT_SIGMOID3])
        sp = native.stream_ptr()
        xp = ptr(x)
        t_fused = timeit(lambda: lib.rovit_kan_stack_fwd(xp, a_w, a_k, a_lw, a_lb, a_o, B, dims, nks, acts, n, sp), 50)
        t_mfma = None
        if all(p[2] is not None for p in prep):
            a_wm = ptr_array([p[2] for p in prep])
            outs_m = [torch.empty_like(o) for o in outs]
            a_om = ptr_array(outs_m)
            t_mfma = timeit(lambda: lib.rovit_kan_stack_fwd_mfma(xp, a_wm, a_k, a_lb, a_om, B, dims, nks, acts, n, sp), 50)
            lib.rovit_kan_stack_fwd(xp, a_w, a_k, a_lw, a_lb, a_o, B, dims, nks, acts, n, sp)
            torch.cuda.synchronize()
            mfma_err = [float((a - b).abs().max()) for a, b in zip(outs, outs_m)]
        ins = [x] + outs[:-1]
        raw = [(ptr(ins[i]), ptr(l.spline_weights), ptr(l.knots), ptr(l.linear.weight), ptr(l.linear.bias), ptr(outs[i]), l.in_features,
                l.out_features, l.knots.numel(), ACT_SIGMOID3 if i == n - 1 else ACT_RELU) for i, l in enumerate(m.kan_layers)]

        def per_layer():
            for (xi, w, k, lw, lb, o, fi, fo, nk, act) in raw:
                lib.rovit_kan_layer_fwd(xi, w, k, lw, lb, o, B, fi, fo, nk, act, sp)
        t_layers = timeit(per_layer, 50)
        # backward kernels only (direct C-ABI calls on the forward's activations, gradients preallocated)
        gouts = [torch.randn_like(o) for o in outs]
        dxs = [torch.empty_like(t) for t in ins]
        dws = [(torch.empty_like(l.spline_weights), torch.empty_like(l.linear.weight), torch.empty_like(l.linear.bias)) for l in m.kan_layers]
        braw = [(ptr(ins[i]), ptr(l.spline_weights), ptr(l.knots), ptr(l.linear.weight), ptr(outs[i]), ptr(gouts[i]), ptr(dxs[i]),
                 ptr(dws[i][0]), ptr(dws[i][1]), ptr(dws[i][2]), B, l.in_features, l.out_features, l.knots.numel(),
                 ACT_SIGMOID3 if i == n - 1 else ACT_RELU, 0, sp) for i, l in enumerate(m.kan_layers)]

        def per_layer_bwd():
            for r in reversed(braw):
                lib.rovit_kan_layer_bwd(*r)
        t_bwd = timeit(per_layer_bwd, 20)
        xg = x.clone().requires_grad_(True)

        def fb():
            for p in m.parameters():
                p.grad = None
            m(xg).sum().backward()
        t_fb = timeit(fb, 10)
        alg = kan_bytes(layers, nb, B)
        print(json.dumps({'shape': name, 'layers': layers, 'num_knots': G, 'batch': B, 'fused_fwd_us': round(t_fused, 1),
                          'per_layer_fwd_us': round(t_layers, 1), 'per_layer_bwd_kernels_us': round(t_bwd, 1), 'fwd_bwd_us': round(t_fb, 1), 'algorithmic_bytes_fwd': alg,
                          'fused_fwd_GBps': round(alg / t_fused / 1e3, 1), 'per_layer_fwd_GBps': round(kan_bytes(layers, nb, B, False) / t_layers / 1e3, 1),
                          'speedup_fwd': round(t_layers / t_fused, 2),
                          'mfma_fwd_us': None if t_mfma is None else round(t_mfma, 1),
                          'mfma_fwd_GBps': None if t_mfma is None else round(alg / t_mfma / 1e3, 1),
                          'mfma_vs_valu_max_abs_diff_per_layer': None if t_mfma is None else mfma_err}), flush=True)


if __name__ == '__main__':
    main()
