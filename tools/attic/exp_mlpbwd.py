import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
M = 256 * 197; sp = native.stream_ptr()
dY = torch.randn(M, 192, device=dev).to(bf); H = torch.randn(M, 192, device=dev).to(bf)
W2T = (torch.randn(768, 192, device=dev) * 0.05).to(bf); W1 = (torch.randn(768, 192, device=dev) * 0.08).to(bf); b1 = torch.randn(768, device=dev) * 0.1
out = torch.empty(M, 768, device=dev, dtype=bf)
t = timeit(lambda: native.call('rovit_gemm_mlp_bwd', native.ptr(dY), 192, native.ptr(H), 192, native.ptr(W2T), native.ptr(W1), native.ptr(b1), M, native.ptr(out), 768, sp), 30)
print(f'mlp_bwd variant {os.environ.get("ROVIT_MLPBWD", "0")}: {t:.1f} us  ({2.0 * (2 * M * 192 + M * 768) / t / 1e6:.2f} TB/s algorithmic)')
mask = torch.rand(M, 768, device=dev).to(bf)
t = timeit(lambda: native.call('rovit_gemm_nt', native.ptr(dY), 192, native.ptr(W2T), 192, M, 768, 192, None, 3, native.ptr(out), 768, None, None, 0, native.ptr(mask), 768, None, 0, sp), 30)
print(f'old fc2 dgrad x stored gelu\': {t:.1f} us')
act = torch.empty(M, 768, device=dev, dtype=bf); dact = torch.empty(M, 768, device=dev, dtype=bf)
for o2 in (None, dact):
    t = timeit(lambda: native.call('rovit_gemm_nt', native.ptr(H), 192, native.ptr(W1), 192, M, 768, 192, native.ptr(b1), 1, native.ptr(act), 768, native.ptr(o2), None, 0, None, 0, None, 0, sp), 30)
    print(f'fc1 fwd gelu, out2={"yes" if o2 is not None else "no"}: {t:.1f} us')
