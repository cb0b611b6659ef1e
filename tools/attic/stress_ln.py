import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from rovit_hip.native import call, ptr
dev = torch.device('cuda:0'); bf = torch.bfloat16
sp = native.stream_ptr()
B, T, D = 256, 197, 192
M = B * T
K = int(sys.argv[1]) if len(sys.argv) > 1 else 192
reps = 300
torch.manual_seed(0)
o = torch.randn(M, K, device=dev).to(bf)
W = (torch.randn(D, K, device=dev) * 0.05).to(bf); b = torch.randn(D, device=dev) * 0.1
X0 = torch.randn(M, D, device=dev)
ref = None
nbad = 0
if len(sys.argv) > 2: call('rovit_set_gemm_debug', int(sys.argv[2]))
for r in range(reps):
    X = X0.clone()
    xh = torch.full((M, D), 77.0, device=dev, dtype=bf); rs = torch.full((M,), -5.0, device=dev)
    call('rovit_gemm_resid_ln', ptr(o), K, ptr(W), K, M, K, ptr(b), ptr(X), ptr(xh), ptr(rs), 1e-6, sp)
    if ref is None:
        ref = (X.clone(), xh.clone(), rs.clone())
        continue
    if not (torch.equal(X, ref[0]) and torch.equal(xh, ref[1]) and torch.equal(rs, ref[2])):
        nbad += 1
        rows = (rs != ref[2]).nonzero().flatten().tolist()
        rowsx = (xh != ref[1]).any(1).nonzero().flatten().tolist()
        rowsX = (X != ref[0]).any(1).nonzero().flatten().tolist()
        print(f'rep {r}: rstd rows {rows[:5]} xhat rows {rowsx[:5]} X rows {rowsX[:5]}')
        for m in rows[:2]:
            lnx0 = 1.0 / torch.sqrt(X0[m].var(unbiased=False) + 1e-6)
            lnx = 1.0 / torch.sqrt(X[m].var(unbiased=False) + 1e-6)
            print(f'   row {m} (tile row {m % (64 if K == 192 else 32)}): rstd got {float(rs[m]):.6f} ref {float(ref[2][m]):.6f}  LN(X0) {float(lnx0):.6f} LN(Xnew) {float(lnx):.6f}; xhat got[:4] {xh[m,:4].tolist()} ref[:4] {ref[1][m,:4].tolist()}')
print('bad reps', nbad, 'of', reps - 1)
