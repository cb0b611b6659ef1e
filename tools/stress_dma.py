"""Race screen for the LDS-DMA GEMM: full-size exact integer GEMM repeated, every element checked."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
dev = torch.device('cuda:0'); bf = torch.bfloat16
sp = native.stream_ptr()
bad = 0
for M, N in ((50432, 576), (50432, 768), (50432, 192), (12345, 576), (1576, 192)):
    K = 192
    for rep in range(6):
        torch.manual_seed(rep)
        A = torch.randint(-3, 4, (M, K), device=dev).float().to(bf)
        W = torch.randint(-2, 3, (N, K), device=dev).float().to(bf)
        out = torch.full((M, N), 7.0, device=dev, dtype=bf)
        # a competing stream of memory traffic to perturb timing
        junk = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
        junk.fill_(rep)
        native.call('rovit_gemm_nt', native.ptr(A), K, native.ptr(W), K, M, N, K, None, 0, native.ptr(out), N, None, None, 0, None, 0, None, 0, sp)
        ref = (A.float() @ W.float().t()).to(bf)
        nbad = int((out != ref).sum())
        bad += nbad
        if nbad:
            rows = (out != ref).any(1).nonzero().flatten()[:8].tolist()
            print('MISMATCH', M, N, rep, nbad, rows)
print('total mismatches', bad)
