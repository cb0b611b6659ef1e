"""Timing ablations of the fused MLP kernels (lockstep 8-wave variant): which part of the launch costs what.
python tools/mlp_ablate.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch
from rovit_hip import native
lib = native.load(); p, sp = native.ptr, native.stream_ptr()
dev = torch.device('cuda:0'); bf = torch.bfloat16
M = 256 * 197
xhat2 = torch.randn(M, 192, device=dev).to(bf)
w1 = (torch.randn(768, 192, device=dev) * 0.08).to(bf); w2 = (torch.randn(192, 768, device=dev) * 0.05).to(bf)
b1, b2 = torch.randn(768, device=dev) * 0.3, torch.randn(192, device=dev) * 0.3
X = torch.randn(M, 192, device=dev); act = torch.empty(M, 768, device=dev, dtype=bf); dact = torch.empty_like(act)
xhat = torch.empty(M, 192, device=dev, dtype=bf); rstd = torch.empty(M, device=dev)
ws = torch.empty(lib.rovit_mlp_stream_bytes(), dtype=torch.uint8, device=dev)
native.call('rovit_mlp_prepare_stream', p(w1), p(w2), p(ws), sp)
st = torch.cuda.current_stream(dev)
def timed(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): fn()
    e1.record(st); e1.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 2)
a = torch.randn(4096, 4096, device=dev, dtype=bf)
for _ in range(50): a @ a
train = lambda: lib.rovit_mlp_fused_fwd(p(xhat2), p(ws), p(b1), p(b2), p(act), p(dact), p(X), p(xhat), p(rstd), 1e-6, M, M, sp)
infer = lambda: lib.rovit_mlp_fused_fwd(p(xhat2), p(ws), p(b1), p(b2), None, None, p(X), p(xhat), p(rstd), 1e-6, M, M, sp)
out = {}
for wv, tag in ((8, 'lockstep'), (9, 'staggered')):
  lib.rovit_set_mlp_waves(wv)
  for bits, name in ((0, 'full'), (1, 'no_epilogue'), (2, 'no_gelu'), (3, 'no_epilogue_no_gelu'), (4, 'no_fc2'), (8, 'no_fc1'), (12, 'no_mfma'), (15, 'skeleton: DMA ring + barriers only'), (14, 'epilogue + ring only')):
    lib.rovit_set_mlp_debug(bits)
    out[tag + ': ' + name] = {'train': timed(train), 'inference': timed(infer)}
lib.rovit_set_mlp_debug(0); lib.rovit_set_mlp_waves(8)
print(json.dumps(out, indent=1))
