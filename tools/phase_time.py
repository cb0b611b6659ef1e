"""GPU wall time of the phases of one training step (device events at the phase boundaries, averaged over steps), next to
the HOST time at which each phase had been enqueued: tells where the device waits for Python and where for kernels.
Developer tool: python tools/phase_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.rovit_kan import RoViTKAN
from rovit_hip.losses import JointLoss
from rovit_hip.optim import RoViTAdamW
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = RoViTKAN(pretrained=False).to(dev).train()
opt = RoViTAdamW(model, lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0)
loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0, torch.ones(4, device=dev))
images = torch.randn(256, 3, 224, 224, device=dev)
labels = torch.randint(0, 4, (256,), device=dev)
names = ['backbone fwd', 'heads+kan fwd', 'loss', 'heads+kan bwd + backbone bwd', 'optimizer']
N = 20
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(8)] for _ in range(N)]
cur = {'e': None}
host = [[0.0] * 6 for _ in range(N)]
st = torch.cuda.current_stream()


def _after_backbone(mod, inp, outp):
    cur['e'][6].record(st)                                 # backbone forward enqueued
    if outp.requires_grad:
        outp.register_hook(lambda g: cur['e'][7].record(st))   # gradient w.r.t. the features is ready: heads/KAN backward done


model.backbone.register_forward_hook(_after_backbone)


def run(e, h):
    cur['e'] = e
    e[0].record(st); h[0] = time.perf_counter()
    out = model(images)
    e[2].record(st); h[2] = time.perf_counter()
    loss = loss_fn(out, labels, labels, 4)['total_loss']
    e[3].record(st); h[3] = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    e[4].record(st); h[4] = time.perf_counter()
    opt.step()
    e[5].record(st); h[5] = time.perf_counter()


for _ in range(5):
    run(ev[0], host[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    run(ev[i], host[i])
torch.cuda.synchronize()
print(f'step {(time.perf_counter() - t0) / N * 1e3:.3f} ms')
seg = [('model forward (backbone + heads + kan)', 0, 2), ('loss', 2, 3), ('backward (heads, kan, backbone)', 3, 4), ('optimizer', 4, 5)]
for name, a, b in seg:
    g = sum(ev[i][a].elapsed_time(ev[i][b]) for i in range(3, N)) / (N - 3)
    hh = sum(host[i][b] - host[i][a] for i in range(3, N)) / (N - 3) * 1e3
    print(f'{name:42s} device {g:7.3f} ms   host enqueue {hh:7.3f} ms')
for name, a, b in (('  backbone forward', 0, 6), ('  heads + KAN forward', 6, 2), ('  heads + KAN backward', 3, 7), ('  backbone backward', 7, 4)):
    g = sum(ev[i][a].elapsed_time(ev[i][b]) for i in range(3, N)) / (N - 3)
    print(f'{name:42s} device {g:7.3f} ms')
g = sum(ev[i][5].elapsed_time(ev[i + 1][0]) for i in range(3, N - 1)) / (N - 4)
print(f'{"between steps":42s} device {g:7.3f} ms')
