import json, os, sys
ROOT = '/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch
from rovit_hip import native
lib = native.load(); p, sp = native.ptr, native.stream_ptr()
dev = torch.device('cuda:0'); bf = torch.bfloat16
T, H = 197, 3
st = torch.cuda.current_stream(dev)
def timed(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
a = torch.randn(4096, 4096, device=dev, dtype=bf)
for _ in range(50): a @ a
out = {}
for B in (85, 171, 256, 512):
    M = B * T
    qkv = torch.randn(M, 576, device=dev).to(bf); o = torch.empty(M, 192, device=dev, dtype=bf)
    lse = torch.empty(B, H, T, device=dev); dO = torch.randn(M, 192, device=dev).to(bf); dqkv = torch.empty(M, 576, device=dev, dtype=bf)
    lib.rovit_attention_fwd(p(qkv), p(o), p(lse), B, T, H, 64, 0.125, sp)
    bwd = lambda: lib.rovit_attention_bwd(p(qkv), p(o), p(lse), p(dO), p(dqkv), B, T, H, 64, 0.125, sp)
    r = {}
    for bits, name in ((0, 'full'), (15, 'empty'), (12, 'compute_only'), (3, 'no_passes')):
        lib.rovit_set_attn_debug(bits); r[name] = round(timed(bwd), 2)
    lib.rovit_set_attn_debug(0); lib.rovit_set_attn_bwd_pipe(0); r['staged'] = round(timed(bwd), 2); lib.rovit_set_attn_bwd_pipe(1)
    out[B] = r
print(json.dumps(out))
