"""Inference forward time at small batches (eval, bf16, stage 4): python tools/fwd_small_batch.py 1 2 4 8 16 32 64 128
Run once per setting of ROVIT_MLP_FUSED / ROVIT_MLP_FUSED_MIN_ROWS to compare the fused MLP half with the two-launch path."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.rovit_kan import RoViTKAN
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = RoViTKAN(pretrained=False).to(dev).eval()
out = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(10):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            m(x)
        torch.cuda.synchronize()
    out[B] = round((time.perf_counter() - t0) / 100 * 1e3, 3)
print(json.dumps({'fused': os.environ.get('ROVIT_MLP_FUSED', '1'), 'ms_per_forward': out}))
