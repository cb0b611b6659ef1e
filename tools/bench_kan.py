"""KAN stack (192 -> 64 -> 16 -> 1) forward and forward+backward timing with algorithmic-byte rates (SURVEY.md 8(d):
activations B*(192+64+64+16+16+1)*4 B forward, weights read once per launch; backward ~2x).  Developer tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.kan import KANSeverityModule
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for B, G in ((256, 5), (512, 32), (16384, 5), (65536, 5), (65536, 32)):
    m = KANSeverityModule([192, 64, 16, 1], num_knots=G, degree=3).to(dev)
    nw = sum(p.numel() for p in m.parameters())
    x = torch.randn(B, 192, device=dev, requires_grad=True)
    def fwd():
        with torch.no_grad():
            m(x)
    def fwdbwd():
        for p in m.parameters(): p.grad = None
        x.grad = None
        m(x).sum().backward()
    tf = timeit(fwd, 20); tb = timeit(fwdbwd, 10)
    act = B * (192 + 64 + 64 + 16 + 16 + 1) * 4
    fbytes = act + nw * 4
    bbytes = 3 * act + 3 * nw * 4
    print(f'B={B:6d} G={G:2d}: fwd {tf:8.1f} us ({fbytes / tf / 1e3:7.1f} GB/s algorithmic)   fwd+bwd {tb:8.1f} us ({bbytes / tb / 1e3:7.1f} GB/s)', flush=True)
