"""Practical HBM ceilings on this box (copy / read-only / write-only) for sizes the GEMM kernels move."""
import torch
dev = torch.device('cuda:0')
def t(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (40, 80, 160, 640, 2560):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum()); tw = t(lambda: b.fill_(1.0))
    ab = a.bfloat16(); bb = torch.empty_like(ab)
    tcb = t(lambda: bb.copy_(ab))
    print(f'{mb:5d} MB  copy {2*n*4/tc/1e12:5.2f} TB/s  read {n*4/tr/1e12:5.2f}  write {n*4/tw/1e12:5.2f}  bf16copy({mb//2}MB) {2*n*2/tcb/1e12:5.2f}', flush=True)
