"""Race screen for the two-stream schedules: repeat forward (+backward) at several batch sizes and compare every
repetition bit for bit with the first one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.backbone import DeiTTiny
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DeiTTiny(12).to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for B in (17, 32, 64, 256):
    x = torch.randn(B, 3, 224, 224, device=dev)
    w = torch.randn(B, 192, device=dev)
    ref_f = ref_i = None; ref_g = None
    nf = ni = ng = 0
    for r in range(reps):
        with torch.no_grad():
            fi = m(x).clone()
        for p in m.parameters(): p.grad = None
        f = m(x)
        (f * w).sum().backward()
        g = torch.cat([p.grad.flatten() for p in m.parameters()])
        if ref_f is None:
            ref_f, ref_i, ref_g = f.detach().clone(), fi, g.clone()
        else:
            nf += int(not torch.equal(f.detach(), ref_f)); ni += int(not torch.equal(fi, ref_i)); ng += int(not torch.equal(g, ref_g))
    print(f'B={B}: {reps} reps, mismatches fwd(train) {nf} fwd(inference) {ni} grads {ng}; inf==train {torch.equal(ref_f, ref_i)}', flush=True)
    bad += nf + ni + ng
print('TOTAL MISMATCHES', bad)
