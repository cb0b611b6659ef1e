#!/bin/bash
# A/B of the attention forward: two workgroups per CU (attn_fwd_kernel) against three (attn_fwd3_kernel: unpadded swizzled tiles, scores computed
# twice), developer library knob 20: bit-identity, stand-alone time, LDS bank conflicts, whole steps
L=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib/librovit_hip_dev.so
ROVIT_HIP_LIB=$L python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
from rovit_hip import native
lib = native.load()
dev = torch.device('cuda:0')
for B, T in ((256, 197), (7, 197), (3, 64), (2, 208), (5, 9)):
    qkv = (torch.randn(B * T, 576, device=dev) * 1.5).to(torch.bfloat16)
    outs = []
    for knob in (0, 1):
        lib.rovit_dev_set_knob(20, knob, 0)
        out = torch.full((B * T, 192), float('nan'), device=dev, dtype=torch.bfloat16); lse = torch.full((B, 3, T), float('nan'), device=dev)
        native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, 3, 64, 0.125, native.stream_ptr())
        outs.append((out, lse))
    torch.cuda.synchronize()
    print('B %d T %d  out identical %s  lse identical %s  nan %s' % (B, T, bool(torch.equal(outs[0][0], outs[1][0])), bool(torch.equal(outs[0][1], outs[1][1])),
          bool(torch.isnan(outs[1][0].float()).any())))
B, T = 256, 197
qkv = (torch.randn(B * T, 576, device=dev) * 1.5).to(torch.bfloat16)
out = torch.empty(B * T, 192, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, 3, T, device=dev)
for knob in (0, 1, 0, 1):
    lib.rovit_dev_set_knob(20, knob, 0)
    fn = lambda: native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, 3, 64, 0.125, native.stream_ptr())
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    print('knob 20 = %d: %.2f us per launch (events around every launch)' % (knob, sum(a.elapsed_time(b) for a, b in ev) / len(ev) * 1e3))
PY
bash tools/ab_knob.sh 20 0 1
