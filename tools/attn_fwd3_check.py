import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
from rovit_hip import native
lib = native.load(); dev = torch.device('cuda:0')
B, T = 4, 197
torch.manual_seed(0)
qkv = (torch.randn(B * T, 576, device=dev) * 1.5).to(torch.bfloat16)
res = []
for knob in (0, 1):
    lib.rovit_dev_set_knob(20, knob, 0)
    out = torch.full((B * T, 192), float('nan'), device=dev, dtype=torch.bfloat16); lse = torch.full((B, 3, T), float('nan'), device=dev)
    native.call('rovit_attention_fwd', native.ptr(qkv), native.ptr(out), native.ptr(lse), B, T, 3, 64, 0.125, native.stream_ptr())
    res.append((out.float(), lse))
x = qkv.float().view(B, T, 3, 3, 64)
q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, 192)
print('old vs ref', float((res[0][0] - ref).abs().max()), ' new vs ref', float((res[1][0] - ref).abs().max()), ' new vs old', float((res[1][0] - res[0][0]).abs().max()),
      ' lse diff', float((res[1][1] - res[0][1]).abs().max()))
d = (res[1][0] - res[0][0]).abs().view(B, T, 3, 64)
print('diff by query row (first image, head 0):', [round(float(d[0, t, 0].max()), 4) for t in (0, 1, 15, 16, 31, 32, 100, 191, 192, 196)])
