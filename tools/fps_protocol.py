"""The reference's `fps()` protocol (evaluation/metrics.py:63-93: batch 1, 10 warm-up, 100 timed forwards, wall clock
with a device sync) on the HIP path, next to its published 2.6 img/s (CPU, KAN active) / 36.7 img/s figures.
Also prints batch-256 inference throughput.  Developer tool: python tools/fps_protocol.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from models.rovit_kan import RoViTKAN
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = RoViTKAN(pretrained=False).to(dev).eval()
for prec, B, warm, n in (('bf16', 1, 10, 100), ('bf16', 256, 5, 30), ('fp32', 1, 10, 100), ('fp32', 256, 3, 10)):
    m.backbone.model.precision = prec
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(warm):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{prec} batch {B}: {n * B / dt:9.1f} images/s  ({dt / n * 1e3:.3f} ms per forward, stage 4, eval)')

# the same forward captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed: batch-1 inference is launch-bound
# (~100 launches), a replay submits them as one graph
for prec, B, n in (('bf16', 1, 100), ('bf16', 8, 100)):
    m.backbone.model.precision = prec
    xs = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                ref = m(xs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(xs)
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(out[k], ref[k]) for k in ('cls_logits', 'kan_severity'))
    print(f'{prec} batch {B} graph replay: {n * B / dt:9.1f} images/s  ({dt / n * 1e3:.3f} ms per forward); outputs identical to the eager call: {same}')
