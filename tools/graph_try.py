import os, sys, time
ROOT = '/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
from rovit_hip.native import call, ptr, ptr_array, stream_ptr
from models.backbone import DeiTTiny
dev = torch.device('cuda:0')
torch.manual_seed(0)
B = 32
m = DeiTTiny(12).to(dev).train()
eng = m.engine
params = m.ordered_parameters()
x = torch.randn(B, 3, 224, 224, device=dev)
f = m(x); f.square().mean().backward()          # warm-up: streams, events, attributes, workspaces
torch.cuda.synchronize()
eng.prepare(params)
ws = eng.take_ws(B, True, dev)
feats = torch.empty(B, 192, device=dev)
parr = ptr_array(params)
def fwd():
    call('rovit_vit_forward', ptr(x), parr, ptr(eng.prep), ptr(ws), ptr(feats), B, 12, 1, 0, stream_ptr())
fwd(); torch.cuda.synchronize(); ref = feats.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode='relaxed'):
        fwd()
    feats.zero_(); g.replay(); torch.cuda.synchronize()
    print('forward graph ok, max diff', float((feats - ref).abs().max()))
    def t(fn, n=100):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print('eager forward %.3f ms   graph replay %.3f ms' % (t(fwd), t(g.replay)))
    eng.ensure_grads(params)
    garr = ptr_array(eng.grad_views)
    df = torch.randn(B, 192, device=dev)
    def bwd():
        call('rovit_vit_backward', ptr(x), ptr(df), parr, ptr(eng.prep), ptr(ws), garr, B, 12, 11, 0, 0, stream_ptr())
    fwd(); bwd(); torch.cuda.synchronize(); gref = eng.grad_flat.clone()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode='relaxed'):
        bwd()
    eng.grad_flat.zero_(); g.replay(); g2.replay(); torch.cuda.synchronize()
    print('backward graph ok, max diff', float((eng.grad_flat - gref).abs().max()), 'scale', float(gref.abs().max()))
    print('eager backward %.3f ms   graph replay %.3f ms' % (t(bwd), t(g2.replay)))
    both = lambda: (g.replay(), g2.replay())
    print('eager fwd+bwd %.3f ms   graphs %.3f ms' % (t(lambda: (fwd(), bwd())), t(both)))
except Exception as e:
    import traceback; traceback.print_exc()
