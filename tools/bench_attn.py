"""Timing of the attention kernels at the benchmark's shape (256 images, 3 heads, 197 tokens): device events, per launch and
back to back.  Both backward kernels (persistent LDS-DMA pipeline / staged) are timed in the same process.  python tools/bench_attn.py [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch  # noqa: E402
from rovit_hip import native  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T, H = 197, 3
    M = B * T
    dev = torch.device('cuda:0')
    lib = native.load()
    p, sp = native.ptr, native.stream_ptr()
    bf = torch.bfloat16
    qkv = torch.randn(M, 576, device=dev).to(bf)
    o = torch.empty(M, 192, device=dev, dtype=bf)
    lse = torch.empty(B, H, T, device=dev)
    dO = torch.randn(M, 192, device=dev).to(bf)
    dqkv = torch.empty(M, 576, device=dev, dtype=bf)
    lib.rovit_attention_fwd(p(qkv), p(o), p(lse), B, T, H, 64, 0.125, sp)
    st = torch.cuda.current_stream(dev)

    def timed(fn, iters=30, per_launch=True):
        for _ in range(5):
            fn()
        if not per_launch:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(iters):
                fn()
            e1.record(st)
            e1.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record(st)
            fn()
            b.record(st)
        evs[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
    a = torch.randn(4096, 4096, device=dev, dtype=bf)
    for _ in range(50):
        a @ a
    fwd = lambda: lib.rovit_attention_fwd(p(qkv), p(o), p(lse), B, T, H, 64, 0.125, sp)
    bwd = lambda: lib.rovit_attention_bwd(p(qkv), p(o), p(lse), p(dO), p(dqkv), B, T, H, 64, 0.125, sp)
    res = {'batch': B, 'ROVIT_ATTN_BWD_WGS': os.environ.get('ROVIT_ATTN_BWD_WGS', '256')}
    for rep in range(3):
        res.setdefault('fwd_us', []).append(round(timed(fwd), 2))
        for pipe in (2, 1, 0):
            lib.rovit_set_attn_bwd_pipe(pipe)
            name = {0: 'staged', 1: 'pipe', 2: 'ring'}[pipe]
            res.setdefault('bwd_us_' + name, []).append(round(timed(bwd), 2))
            res.setdefault('bwd_b2b_us_' + name, []).append(round(timed(bwd, per_launch=False), 2))
    lib.rovit_set_attn_bwd_pipe(1)          # the ablations below are of the pipelined kernel
    for bits, name in ((1, 'bwd_us_no_pass1'), (2, 'bwd_us_no_pass2'), (3, 'bwd_us_no_passes'), (4, 'bwd_us_no_tile_loads'), (8, 'bwd_us_no_stores'), (12, 'bwd_us_compute_only'), (15, 'bwd_us_empty')):
        lib.rovit_set_attn_debug(bits)
        res[name] = round(timed(bwd), 2)
    lib.rovit_set_attn_debug(0)
    lib.rovit_set_attn_bwd_pipe(0)
    for bits, name in ((1, 'staged_no_pass1'), (2, 'staged_no_pass2'), (3, 'staged_no_passes')):      # the default kernel's own ablations
        lib.rovit_set_attn_debug(bits)
        res[name] = round(timed(bwd), 2)
    lib.rovit_set_attn_debug(0)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
