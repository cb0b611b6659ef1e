"""Run ONE kernel configuration a few times (for rocprofv3 --pmc passes).  usage: one_kernel.py wgrad_fc1|qkv|fc1|attn_fwd|attn_bwd [splits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
dev = torch.device('cuda:0'); bf = torch.bfloat16
lib = native.load()
which = sys.argv[1]
B, T = 256, 197
M = B * T
sp = native.stream_ptr()
if which.startswith('wgrad'):
    N, K = {'wgrad_fc1': (768, 192), 'wgrad_fc2': (192, 768), 'wgrad_qkv': (576, 192), 'wgrad_proj': (192, 192)}[which]
    s = int(sys.argv[2]) if len(sys.argv) > 2 else lib.rovit_wgrad_splits(M, N, K)
    dY = torch.randn(M, N, device=dev).to(bf); A = torch.randn(M, K, device=dev).to(bf)
    ws = torch.empty(lib.rovit_wgrad_workspace_bytes(N, K, s) // 4, device=dev)
    fn = lambda: native.call('rovit_wgrad', native.ptr(dY), N, native.ptr(A), K, M, N, K, s, 0, native.ptr(ws), sp)
elif which in ('qkv', 'fc1', 'fc2', 'fc1d'):
    N, K, epi = {'qkv': (576, 192, 0), 'fc1': (768, 192, 1), 'fc2': (192, 768, 2), 'fc1d': (192, 768, 0)}[which]
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.05).to(bf); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=bf); out2 = torch.empty(M, N, device=dev, dtype=bf); xres = torch.zeros(M, N, device=dev)
    fn = lambda: native.call('rovit_gemm_nt', native.ptr(A), K, native.ptr(W), K, M, N, K, native.ptr(bias), epi, native.ptr(out), N,
                             native.ptr(out2) if epi == 1 else None, native.ptr(xres) if epi == 2 else None, N, None, 0, None, 0, sp)
else:
    qkv = torch.randn(M, 576, device=dev).to(bf); out = torch.empty(M, 192, device=dev, dtype=bf); lse = torch.empty(B, 3, T, device=dev)
    dout = torch.randn(M, 192, device=dev).to(bf); dqkv = torch.empty_like(qkv)
    native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, 3, 64, 0.125, sp)
    if which == 'attn_fwd':
        fn = lambda: native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, 3, 64, 0.125, sp)
    else:
        fn = lambda: native.call('rovit_attention_bwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), native.ptr(dout), native.ptr(dqkv), B, T, 3, 64, 0.125, sp)
for _ in range(5):
    fn()
torch.cuda.synchronize()
