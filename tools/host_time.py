"""How long does the HOST need to enqueue one training step (no device sync inside the loop)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.rovit_kan import RoViTKAN
from rovit_hip.losses import JointLoss
from rovit_hip.optim import RoViTAdamW
dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = RoViTKAN(pretrained=False).to(dev).train()
opt = RoViTAdamW(m)
lf = JointLoss(1.0, 0.5, 0.5, 2.0, torch.ones(4, device=dev))
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 4, (B,), device=dev)
def step():
    out = m(x); loss = lf(out, y, y, 4)['total_loss']; opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(20): step()
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
print(f'B={B}: host enqueue {t_host*1e3:.2f} ms/step, wall {t_all*1e3:.2f} ms/step')
if B <= 8:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
