import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from oracle import ref_cpu
from models.rovit_kan import RoViTKAN
from rovit_hip import taps
dev = torch.device('cuda:0')
sd = ref_cpu.init_rovit_state(depth=12, seed=17)
m = RoViTKAN(pretrained=False); m.load_state_dict(sd); m = m.to(dev).eval()
x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(3))
cap = {}
t = m.backbone.model.blocks[-1].norm1
h1 = t.register_forward_hook(lambda mod, i, o: cap.__setitem__('act', o.detach()))
h2 = t.register_full_backward_hook(lambda mod, gi, go: cap.__setitem__('grad', go[0].detach()))
out = m(x.to(dev).requires_grad_(True))
cls = int(out['cls_logits'].argmax(1))
eng = m.backbone.model.engine
ws, B = eng.last_ws
out['cls_logits'][0, cls].backward()
tp = {}
feats = ref_cpu.vit_forward(x, sd, prefix='backbone.model.', tap_norm1=(11, tp))
logits = ref_cpu.heads_forward(feats, sd, 1)['cls_logits']
gref, = torch.autograd.grad(logits[0, cls], tp['y'])
g = cap['grad'].cpu()[0]; r = gref[0]
print('row norms got', g.norm(dim=1)[:6].tolist(), 'ref', r.norm(dim=1)[:6].tolist())
rel = (g - r).norm(dim=1) / r.norm(dim=1).clamp_min(1e-12)
print('per-row rel err: row0 %.4f, median %.4f, max %.4f at %d' % (rel[0], rel.median(), rel.max(), int(rel.argmax())))
print('overall rel', float((g - r).norm() / r.norm()))
# reference dqkv from the oracle: grad wrt qkv output
# isolate the tap GEMM: same dqkv through torch
orig = taps.norm1_output_grad
def patched(params, ws, batch, depth, block):
    out = orig(params, ws, batch, depth, block)
    dqkv = taps.workspace_view(ws, batch, depth, taps.WS_DQKV, block).float()
    w = params[6 + 12 * block + 2].detach()
    t = (dqkv @ w).view(batch, 197, 192)
    print('tap gemm vs torch: rel', float((out - t).norm() / t.norm()), ' sections |dq|,|dk|,|dv|:', float(dqkv[:, :192].norm()), float(dqkv[:, 192:384].norm()), float(dqkv[:, 384:].norm()))
    cap['torch'] = t
    cap['dqkv'] = dqkv.clone()
    return out
taps.norm1_output_grad = patched
h2 = t.register_full_backward_hook(lambda mod, gi, go: cap.__setitem__('grad', go[0].detach()))
out = m(x.to(dev).requires_grad_(True))
out['cls_logits'][0, cls].backward()
r = gref[0]
for nm, g in (('hip tap', cap['grad'].cpu()[0]), ('torch on hip dqkv', cap['torch'].cpu()[0])):
    print(nm, 'overall rel', float((g - r).norm() / r.norm()))
# oracle dqkv: gradient wrt the qkv output of block 11
sd2 = {k: v.clone() for k, v in sd.items()}
tp2 = {}
feats = ref_cpu.vit_forward(x, sd2, prefix='backbone.model.', tap_norm1=(11, tp2))
y = tp2['y']
W = sd['backbone.model.blocks.11.attn.qkv.weight']; bq = sd['backbone.model.blocks.11.attn.qkv.bias']
logit = ref_cpu.heads_forward(feats, sd2, 1)['cls_logits'][0, cls]
gy, = torch.autograd.grad(logit, y)
# dqkv_ref solves gy = dqkv W  (W has full column rank 192 of 576): least squares is not unique -> compare projections instead
proj = cap['dqkv'].cpu() @ W
print('hip dqkv @ W (cpu fp32) vs oracle dY: rel', float((proj - gy[0]).norm() / gy[0].norm()))
# ---- oracle dqkv by hand: tokens entering block 11, then block 11 + final norm + cls head with qkv as a leaf
import torch.nn.functional as F
P = 'backbone.model.'
sd11 = {k: v for k, v in sd.items() if not k.startswith(P + 'blocks.11.')}
with torch.no_grad():
    t_in = ref_cpu.vit_forward(x, sd11, prefix=P, return_tokens=True)
b = P + 'blocks.11.'
hh = F.layer_norm(t_in, (192,), sd[b + 'norm1.weight'], sd[b + 'norm1.bias'], 1e-6)
qkv = F.linear(hh, sd[b + 'attn.qkv.weight'], sd[b + 'attn.qkv.bias']).detach().requires_grad_(True)
q3 = qkv.reshape(1, -1, 3, 3, 64).permute(2, 0, 3, 1, 4)
a = torch.softmax((q3[0] * 0.125) @ q3[1].transpose(-2, -1), dim=-1)
o = (a @ q3[2]).transpose(1, 2).reshape(1, -1, 192)
t1 = t_in + F.linear(o, sd[b + 'attn.proj.weight'], sd[b + 'attn.proj.bias'])
h2_ = F.layer_norm(t1, (192,), sd[b + 'norm2.weight'], sd[b + 'norm2.bias'], 1e-6)
t2 = t1 + F.linear(F.gelu(F.linear(h2_, sd[b + 'mlp.fc1.weight'], sd[b + 'mlp.fc1.bias'])), sd[b + 'mlp.fc2.weight'], sd[b + 'mlp.fc2.bias'])
f = F.layer_norm(t2, (192,), sd[P + 'norm.weight'], sd[P + 'norm.bias'], 1e-6)[:, 0]
lg = ref_cpu.heads_forward(f, sd, 1)['cls_logits'][0, cls]
dq_ref, = torch.autograd.grad(lg, qkv)
dq_ref = dq_ref[0]; dq_hip = cap['dqkv'].cpu()
for s, nm in ((slice(0, 192), 'q'), (slice(192, 384), 'k'), (slice(384, 576), 'v')):
    a_, r_ = dq_hip[:, s], dq_ref[:, s]
    print(f'dqkv {nm}: rel err {float((a_ - r_).norm() / r_.norm()):.4f}; after removing the mean over tokens: {float(((a_ - a_.mean(0)) - (r_ - r_.mean(0))).norm() / (r_ - r_.mean(0)).norm()):.4f}; |mean diff| {float((a_.mean(0) - r_.mean(0)).norm()):.3e} vs |mean ref| {float(r_.mean(0).norm()):.3e}')
print('row 0 of dv: hip', dq_hip[0, 384:388].tolist(), 'ref', dq_ref[0, 384:388].tolist())
print('row 5 of dv: hip', dq_hip[5, 384:388].tolist(), 'ref', dq_ref[5, 384:388].tolist())
