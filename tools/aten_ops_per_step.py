"""Which aten ops (copies, fills, elementwise kernels) a training step of bench.py's shape issues besides the library's own launches:
torch.profiler over 3 steps, ops with a device kernel, grouped by name and input shapes, with the Python frame that issued them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch
from torch.profiler import profile, ProfilerActivity
from models.rovit_kan import RoViTKAN
from rovit_hip.losses import JointLoss
from rovit_hip.parallel import GradSync
from rovit_hip.optim import RoViTAdamW
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = RoViTKAN(pretrained=False).to(dev).train()
model.curriculum_stage = 4
opt = RoViTAdamW(model, lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0)
loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0, torch.ones(4, device=dev))
sync = GradSync(model, buckets=2, force=False, optimizer=opt)
images = torch.randn(256, 3, 224, 224, device=dev)
labels = torch.randint(0, 4, (256,), device=dev)
def step():
    out = model(images)
    loss = loss_fn(out, labels, labels, 4)['total_loss']
    opt.zero_grad(set_to_none=True)
    loss.backward()
    sync.finish()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4):
    if e.device_time_total > 0 and e.key.startswith('aten::'):
        rows.append((e.count / 3, e.key, str(e.input_shapes)[:70], e.device_time_total / 3, ' <- '.join(s.split('/')[-1] for s in e.stack[:3])))
rows.sort(key=lambda r: -r[3])
for r in rows[:40]:
    print('%5.1f/step %-28s %-70s %7.1f us/step  %s' % r)
