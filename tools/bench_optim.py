"""Kernel times of the fused optimizer launches (csrc/optim.hip) alone: rovit_sq_norm_clip over the backbone's 5.5 M gradient floats
(+ the head / KAN buffer) and rovit_adamw_flat_multi.  Developer tool: python tools/bench_optim.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd'))
import torch
from rovit_hip import native
dev = torch.device('cuda:0')
lib = native.load()
n0, n1 = 5524416, 182000
g = [torch.randn(n0, device=dev), torch.randn(n1, device=dev)]
p = [torch.randn_like(t) for t in g]; m = [torch.zeros_like(t) for t in g]; v = [torch.zeros_like(t) for t in g]
coef = torch.ones((), device=dev); norm = torch.zeros((), device=dev); scratch = torch.zeros(2048, device=dev)
st = torch.cuda.current_stream().cuda_stream
arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
cnt = (C.c_size_t * 2)(n0, n1)


def timed(fn, n=200, flush=None):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)                      # 512 MB through the caches: the gradients come from HBM, as in the step
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


norm_fn = lambda: native.check(lib.rovit_sq_norm_clip(arr(g), cnt, 2, 1.0, coef.data_ptr(), norm.data_ptr(), scratch.data_ptr(), scratch.numel(), st), 'norm')
lr = (C.c_float * 2)(1e-5, 1e-4); tt = (C.c_int * 2)(3, 3)
adam_fn = lambda: native.check(lib.rovit_adamw_flat_multi(arr(p), arr(g), arr(m), arr(v), cnt, lr, tt, 2, coef.data_ptr(), 0.9, 0.999, 1e-8, 1e-4, st), 'adam')
big = torch.zeros(128 * 1024 * 1024, device=dev)
print('sq_norm_clip   warm %6.2f us   cold %6.2f us   (22.8 MB read)' % (timed(norm_fn), timed(norm_fn, 30, big)))
print('adamw_multi    warm %6.2f us   cold %6.2f us   (160 MB moved)' % (timed(adam_fn), timed(adam_fn, 30, big)))
ref = float(torch.sqrt(g[0].double().square().sum() + g[1].double().square().sum()))
print('norm %.4f (torch fp64 %.4f)' % (float(norm), ref))
