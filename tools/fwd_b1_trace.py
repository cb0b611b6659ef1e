"""Batch-1 inference forwards in a loop (for rocprofv3 --kernel-trace): python tools/fwd_b1_trace.py [batch] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.rovit_kan import RoViTKAN
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = RoViTKAN(pretrained=False).to(dev).eval()
x = torch.randn(B, 3, 224, 224, device=dev)
with torch.no_grad():
    for _ in range(n):
        m(x)
torch.cuda.synchronize()
