import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
M = 256 * 197
for name, N, K, epi in (('qkv', 576, 192, 0), ('fc1 gelu', 768, 192, 1), ('proj dgrad', 192, 192, 0)):
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.05).to(bf); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=bf); out2 = torch.empty(M, N, device=dev, dtype=bf)
    sp = native.stream_ptr()
    def run():
        native.call('rovit_gemm_nt', native.ptr(A), K, native.ptr(W), K, M, N, K, native.ptr(bias), epi, native.ptr(out), N, native.ptr(out2) if epi == 1 else None,
                    None, N, None, N, None, 0, sp)
    r = {}
    for d in (0, 1, 2, 4, 8, 3, 7, 15):
        native.call('rovit_set_gemm_debug', d)
        r[d] = timeit(run, 20)
    native.call('rovit_set_gemm_debug', 0)
    print(f'{name:10s} ' + '  '.join(f'dbg{d}:{v:5.1f}' for d, v in r.items()) + '   (1=no store 2=no mfma 4=no dma 8=no gelu)', flush=True)
