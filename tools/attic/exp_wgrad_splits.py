"""M-split sweep of the merged (per-block) weight-gradient launch at the headline shapes (developer tool)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from tools.bench_kernels import timeit
dev = torch.device('cuda:0'); bf = torch.bfloat16
lib = native.load()
M = 256 * 197
shapes = [(576, 192), (192, 768), (768, 192), (192, 192)]
dY = [torch.randn(M, n, device=dev).to(bf) for n, _ in shapes]
A = [torch.randn(M, k, device=dev).to(bf) for _, k in shapes]
arr = lambda xs: (C.c_int * len(xs))(*xs)
a_dy, a_a = native.ptr_array(dY), native.ptr_array(A)
ldy, lda, Ns, Ks = arr([n for n, _ in shapes]), arr([k for _, k in shapes]), arr([n for n, _ in shapes]), arr([k for _, k in shapes])
alg = sum(2.0 * M * (n + k) for n, k in shapes)
a = torch.randn(4096, 4096, device=dev, dtype=bf)
for _ in range(200):
    a @ a
torch.cuda.synchronize()
for splits in [int(x) for x in (sys.argv[1:] or ('8', '10', '11', '12', '14', '16', '18', '20', '21', '22', '24', '28', '32'))]:
    ws = [torch.empty(lib.rovit_wgrad_workspace_bytes(n, k, splits), dtype=torch.uint8, device=dev) for n, k in shapes]
    a_ws = native.ptr_array(ws)
    t = timeit(lambda: native.call('rovit_wgrad_multi', a_dy, ldy, a_a, lda, Ns, Ks, a_ws, 4, M, splits, native.stream_ptr()), 40)
    slab = sum(4.0 * n * k * splits for n, k in shapes)
    print(f'splits {splits:3d}  wgs {24 * splits:4d}  {t:6.1f} us  slab {slab / 1e6:5.1f} MB  alg {(alg + slab) / t / 1e6:5.2f} TB/s', flush=True)
