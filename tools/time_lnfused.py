"""Isolated timing of the LayerNorm-fused GEMMs at the headline size (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from rovit_hip.native import call, ptr
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
sp = native.stream_ptr()
M, D = 256 * 197, 192
for K in (192, 576, 768):
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(D, K, device=dev) * 0.05).to(bf); b = torch.randn(D, device=dev)
    X = torch.randn(M, D, device=dev); xh = torch.empty(M, D, device=dev, dtype=bf); rs = torch.rand(M, device=dev) + 0.5
    xb = torch.empty(M, D, device=dev, dtype=bf)
    t1 = timeit(lambda: call('rovit_gemm_resid_ln', ptr(A), K, ptr(W), K, M, K, ptr(b), ptr(X), ptr(xh), ptr(rs), 1e-6, sp), 30)
    t2 = timeit(lambda: call('rovit_gemm_ln_bwd', ptr(A), K, ptr(W), K, M, K, ptr(xh), ptr(rs), ptr(X), None, ptr(xb), sp), 30)
    print(f'K={K}: resid+LN {t1:6.1f} us   dgrad+LN-bwd {t2:6.1f} us', flush=True)
