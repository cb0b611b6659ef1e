"""How much the benchmark's 25-step loss trajectory moves under changes that are all 'the same arithmetic to bf16 rounding': the one-launch
against the two-launch MLP half (fp32 summation order of fc2), and -- with ROVIT_HIP_LIB pointing at another build -- the residual gradient
stored as bf16 rows (round 4) against fp32 (before).  Same seeds, same dropout masks, same data as bench.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch  # noqa: E402
from rovit_hip import native  # noqa: E402
from rovit_hip.functions import VitEngine  # noqa: E402
from models.rovit_kan import RoViTKAN  # noqa: E402
from rovit_hip.losses import JointLoss  # noqa: E402
from rovit_hip.optim import RoViTAdamW  # noqa: E402

dev = torch.device('cuda:0')
res = {'library': os.environ.get('ROVIT_HIP_LIB', 'product')}
for name, path in (('auto_one_launch', native.MLP_AUTO), ('two_launch', native.MLP_TWO_LAUNCH)):
    VitEngine.default_mlp_path = path
    torch.manual_seed(0)
    model = RoViTKAN(pretrained=False).to(dev).train()
    model.curriculum_stage = 4
    opt = RoViTAdamW(model, lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0)
    loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0, torch.ones(4, device=dev))
    g = torch.Generator(device=dev).manual_seed(1000)
    images = torch.randn(256, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 4, (256,), device=dev, generator=g)
    losses = []
    for _ in range(25):
        out = model(images)
        loss = loss_fn(out, labels, labels, 4)['total_loss']
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(round(float(loss.detach()), 5))
    res[name] = losses
VitEngine.default_mlp_path = native.MLP_AUTO
print(json.dumps(res))
