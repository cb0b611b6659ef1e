"""Race screen for the dgrad + LayerNorm-backward GEMM at full size: same inputs, many launches, bit-compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from rovit_hip.native import call, ptr
dev = torch.device('cuda:0'); bf = torch.bfloat16
sp = native.stream_ptr()
M, D = 256 * 197, 192
K = int(sys.argv[1]) if len(sys.argv) > 1 else 576
if len(sys.argv) > 2: call('rovit_set_gemm_debug', int(sys.argv[2]))
torch.manual_seed(0)
dY = torch.randn(M, K, device=dev).to(bf)
W = (torch.randn(D, K, device=dev) * 0.05).to(bf)
xh = torch.randn(M, D, device=dev).to(bf); rstd = torch.rand(M, device=dev) + 0.5
X0 = torch.randn(M, D, device=dev)
ref = None; nbad = 0
for r in range(200):
    dX = X0.clone(); dXb = torch.full((M, D), 7.0, device=dev, dtype=bf)
    call('rovit_gemm_ln_bwd', ptr(dY), K, ptr(W), K, M, K, ptr(xh), ptr(rstd), ptr(dX), ptr(dXb), sp)
    if ref is None: ref = (dX.clone(), dXb.clone()); continue
    if not (torch.equal(dX, ref[0]) and torch.equal(dXb, ref[1])):
        nbad += 1
        if nbad <= 3:
            rows = (dX != ref[0]).any(1).nonzero().flatten()
            print('rep', r, 'rows differing', rows.numel(), rows[:8].tolist(), 'tile rows', [int(x) % 32 for x in rows[:8]])
print('bad reps', nbad, 'of 199')
