import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
M = 256 * 197
for name, N, K, epi in (('qkv', 576, 192, 0), ('fc1 gelu', 768, 192, 1), ('fc2 resid', 192, 768, 2), ('fc1 dgrad', 192, 768, 0), ('mul', 768, 192, 3)):
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.05).to(bf); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=bf); out2 = torch.empty(M, N, device=dev, dtype=bf); xres = torch.zeros(M, N, device=dev)
    mul = torch.rand(M, N, device=dev).to(bf)
    sp = native.stream_ptr()
    def run():
        native.call('rovit_gemm_nt', native.ptr(A), K, native.ptr(W), K, M, N, K, native.ptr(bias), epi, native.ptr(out), N, native.ptr(out2) if epi == 1 else None,
                    native.ptr(xres) if epi == 2 else None, N, native.ptr(mul) if epi == 3 else None, N, None, 0, sp)
    r = []
    for d in (0, 1, 2, 3):
        native.call('rovit_set_gemm_debug', d)
        r.append(timeit(run, 20))
    native.call('rovit_set_gemm_debug', 0)
    print(f'{name:10s} full {r[0]:6.1f}  no-store {r[1]:6.1f}  no-mfma {r[2]:6.1f}  neither {r[3]:6.1f} us', flush=True)
