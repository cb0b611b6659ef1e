"""Soak test: two independent training runs (same seed, batch 256, dropout on, two-stream schedule) must end in
bit-identical parameters after N steps; prints the loss curve.  python tools/soak.py [steps]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.rovit_kan import RoViTKAN
from rovit_hip.losses import JointLoss
from rovit_hip.optim import RoViTAdamW
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(5)
m0 = RoViTKAN(pretrained=False).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev)
y = torch.randint(0, 4, (256,), device=dev)

def run():
    m = copy.deepcopy(m0).train()
    opt = RoViTAdamW(m, lr=5e-4, weight_decay=1e-4, max_grad_norm=1.0)
    lf = JointLoss()
    torch.manual_seed(9)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = lf(m(x), y, y, 4)['total_loss']
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return m, torch.stack(losses)

ma, la = run()
mb, lb = run()
torch.cuda.synchronize()
bad = [n for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()) if not torch.equal(p, q)]
print('loss first/last:', float(la[0]), float(la[-1]), ' finite:', bool(torch.isfinite(la).all()))
print('loss curves identical:', bool(torch.equal(la, lb)), ' parameters differing:', len(bad), bad[:5])
