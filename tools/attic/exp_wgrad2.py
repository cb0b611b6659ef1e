"""Tile / split sweep of the weight-gradient kernel at the headline shapes (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
lib = native.load()
M = 256 * 197; sp = native.stream_ptr()
shapes = (('fc1', 768, 192), ('fc2', 192, 768), ('qkv', 576, 192), ('proj', 192, 192))
tiles = ((96, 96), (64, 96), (96, 64), (64, 64), (32, 96), (96, 32), (96, 192), (192, 96), (64, 192), (192, 64))
for name, N, K in shapes:
    dY = torch.randn(M, N, device=dev).to(bf); A = torch.randn(M, K, device=dev).to(bf)
    ref = None
    for tn, tk in tiles:
        if N % tn or K % tk:
            continue
        native.call('rovit_set_wgrad_tile', tn, tk)
        ntl = (N // tn) * (K // tk)
        for wgs in (256, 512, 768):
            s = max(1, (wgs + ntl - 1) // ntl)
            ws = torch.empty(lib.rovit_wgrad_workspace_bytes(N, K, s) // 4, device=dev)
            dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            t1 = timeit(lambda: native.call('rovit_wgrad', native.ptr(dY), N, native.ptr(A), K, M, N, K, s, 0, native.ptr(ws), sp), 30)
            t2 = timeit(lambda: native.call('rovit_wgrad_reduce', native.ptr(ws), s, N, K, None, None, None, native.ptr(dW), native.ptr(db), None, None, None, sp), 30)
            if ref is None:
                ref = (dW.clone(), db.clone())
            err = float((dW - ref[0]).abs().max() / ref[0].abs().max()); errb = float((db - ref[1]).abs().max() / ref[1].abs().max())
            alg = 2.0 * M * (N + K) + 4.0 * N * K
            print(f'{name} tile {tn:3d}x{tk:3d} wgs {s * ntl:4d} splits {s:3d}: wgrad {t1:6.1f} us ({alg / t1 / 1e6:5.2f} TB/s alg, slab {s * N * K * 4 / 1e6:5.1f} MB)  reduce {t2:5.1f} us  sum {t1 + t2:6.1f}  relerr {err:.1e} {errb:.1e}', flush=True)
native.call('rovit_set_wgrad_tile', 0, 0)
