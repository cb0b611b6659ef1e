import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
lib = native.load()
M = 256 * 197; sp = native.stream_ptr()
for name, N, K in (('fc1', 768, 192), ('proj', 192, 192)):
    dY = torch.randn(M, N, device=dev).to(bf); A = torch.randn(M, K, device=dev).to(bf)
    for s in (16, 32):
        ws = torch.empty(lib.rovit_wgrad_workspace_bytes(N, K, s) // 4, device=dev)
        r = []
        for d in (0, 16, 32, 48):
            native.call('rovit_set_gemm_debug', d)
            r.append(timeit(lambda: native.call('rovit_wgrad', native.ptr(dY), N, native.ptr(A), K, M, N, K, s, 0, native.ptr(ws), sp), 20))
        native.call('rovit_set_gemm_debug', 0)
        print(f'{name} S={s}: full {r[0]:.1f}  no-loads {r[1]:.1f}  no-mfma {r[2]:.1f}  neither {r[3]:.1f} us', flush=True)
