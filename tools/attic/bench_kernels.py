"""Per-kernel timing of the backbone's HIP kernels at the headline shapes (B=256 -> M=50432 rows), with device
events on the launch stream.  Developer tool: prints one line per kernel with achieved TFLOP/s or GB/s.

    python tools/bench_kernels.py [--batch 256] [--iters 20] [--only gemm,wgrad,attn,ln]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from rovit_hip import native  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16


def timeit(fn, iters):
    for _ in range(3):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def gemm_case(name, M, N, K, epi, iters, check=True):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=bf)
    out2 = torch.empty(M, N, device=dev, dtype=bf) if epi == 1 else None
    xres = torch.zeros(M, N, device=dev) if epi == 2 else None
    mul = torch.rand(M, N, device=dev).to(bf) if epi == 3 else None
    sp = native.stream_ptr()

    def run():
        native.call('rovit_gemm_nt', native.ptr(A), K, native.ptr(W), K, M, N, K, native.ptr(bias), epi,
                    native.ptr(out) if epi != 2 else None, N, native.ptr(out2), native.ptr(xres), N, native.ptr(mul), N, None, 0, sp)
    res = []
    for tile in (0, 2, 3):
        native.call('rovit_set_gemm_tile', tile)
        us = timeit(run, iters)
        res.append(us)
    native.call('rovit_set_gemm_tile', 0)
    err = ''
    if check and epi in (0, 3):
        run()
        ref = A[:2048].float() @ W.float().t() + bias
        if epi == 3:
            ref = ref * mul[:2048].float()
        err = ' relerr=%.2e' % float((out[:2048].float() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    print(f'{name:28s} M={M} N={N:4d} K={K:4d} epi={epi}  ws {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF)  '
          f't192 {res[1]:7.1f} us  t96 {res[2]:7.1f} us{err}', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='gemm,wgrad,attn,ln')
    a = ap.parse_args()
    lib = native.load()
    B, T = a.batch, 197
    M = B * T
    sp = native.stream_ptr()
    only = a.only.split(',')
    if 'gemm' in only:
        gemm_case('qkv fwd', M, 576, 192, 0, a.iters)
        gemm_case('proj fwd (resid)', M, 192, 192, 2, a.iters)
        gemm_case('fc1 fwd (gelu)', M, 768, 192, 1, a.iters)
        gemm_case('fc2 fwd (resid)', M, 192, 768, 2, a.iters)
        gemm_case('fc2 dgrad (mul)', M, 768, 192, 3, a.iters)
        gemm_case('fc1 dgrad', M, 192, 768, 0, a.iters)
        gemm_case('proj dgrad', M, 192, 192, 0, a.iters)
        gemm_case('qkv dgrad', M, 192, 576, 0, a.iters)
    if 'wgrad' in only:
        for name, N, K in (('wgrad qkv', 576, 192), ('wgrad proj', 192, 192), ('wgrad fc1', 768, 192), ('wgrad fc2', 192, 768)):
            dY = torch.randn(M, N, device=dev).to(bf)
            A = torch.randn(M, K, device=dev).to(bf)
            s0 = lib.rovit_wgrad_splits(M, N, K)
            for s in (s0, 2 * s0, 4 * s0):
                ws = torch.empty(lib.rovit_wgrad_workspace_bytes(N, K, s) // 4, device=dev)
                dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
                t1 = timeit(lambda: native.call('rovit_wgrad', native.ptr(dY), N, native.ptr(A), K, M, N, K, s, 0, native.ptr(ws), sp), a.iters)
                t2 = timeit(lambda: native.call('rovit_wgrad_reduce', native.ptr(ws), s, N, K, None, None, None, native.ptr(dW), native.ptr(db),
                                                None, None, None, sp), a.iters)
                print(f'{name:28s} splits={s:3d} wgrad {t1:7.1f} us ({2.0 * M * N * K / t1 / 1e6:6.1f} TF)  reduce {t2:6.1f} us', flush=True)
    if 'attn' in only:
        qkv = (torch.randn(M, 576, device=dev)).to(bf)
        out = torch.empty(M, 192, device=dev, dtype=bf)
        lse = torch.empty(B, 3, T, device=dev)
        dout = torch.randn(M, 192, device=dev).to(bf)
        dqkv = torch.empty_like(qkv)
        t1 = timeit(lambda: native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, 3, 64, 0.125, sp), a.iters)
        t2 = timeit(lambda: native.call('rovit_attention_bwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), native.ptr(dout),
                                        native.ptr(dqkv), B, T, 3, 64, 0.125, sp), a.iters)
        fl = 4.0 * B * 3 * T * T * 64
        print(f'attention fwd {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF algorithmic)   bwd {t2:7.1f} us ({2.5 * fl / t2 / 1e6:6.1f} TF)', flush=True)
    if 'ln' in only:
        x = torch.randn(M, 192, device=dev)
        xh = torch.empty(M, 192, device=dev, dtype=bf)
        rs = torch.empty(M, device=dev)
        g = torch.randn(M, 192, device=dev).to(bf)
        dX = torch.zeros(M, 192, device=dev)
        dXb = torch.empty(M, 192, device=dev, dtype=bf)
        t1 = timeit(lambda: native.call('rovit_layernorm_fwd', native.ptr(x), native.ptr(xh), native.ptr(rs), M, 192, 1e-6, sp), a.iters)
        t2 = timeit(lambda: native.call('rovit_layernorm_bwd', native.ptr(g), native.ptr(xh), native.ptr(rs), native.ptr(dX), native.ptr(dXb), M, 192, sp), a.iters)
        print(f'layernorm fwd {t1:6.1f} us ({M * 192 * 6 / t1 / 1e3:6.0f} GB/s)   bwd {t2:6.1f} us ({M * 192 * 14 / t2 / 1e3:6.0f} GB/s)', flush=True)


if __name__ == '__main__':
    main()
