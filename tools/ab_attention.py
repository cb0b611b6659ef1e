"""A/B of the attention kernels in ONE process on one box (developer library: make -C csrc dev): round 3's one-workgroup-per-CU
kernels against round 4's two-workgroups-per-CU kernels, interleaved repetitions, device events per launch and back to back; the two
builds' outputs are compared bit for bit.  ROVIT_HIP_LIB is set here; run as  python tools/ab_attention.py [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')
os.environ.setdefault('ROVIT_HIP_LIB', os.path.join(PKG, 'lib', 'librovit_hip_dev.so'))
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402
from rovit_hip import native  # noqa: E402

KNOB_FWD_R3, KNOB_BWD_R3, KNOB_DBG, KNOB_SPLIT = 0, 1, 2, 3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T, H = 197, 3
    M = B * T
    dev = torch.device('cuda:0')
    lib = native.load()
    assert hasattr(lib, 'rovit_dev_set_knob'), 'needs the developer library (make -C csrc dev)'
    p, sp = native.ptr, native.stream_ptr()
    bf = torch.bfloat16
    qkv = torch.randn(M, 576, device=dev).to(bf)
    o = torch.empty(M, 192, device=dev, dtype=bf)
    lse = torch.empty(B, H, T, device=dev)
    dO = torch.randn(M, 192, device=dev).to(bf)
    dqkv = torch.empty(M, 576, device=dev, dtype=bf)
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

    x = torch.randn(256 * 197, 192, device=dev)
    xh = torch.empty(256 * 197, 192, device=dev, dtype=bf)
    rs = torch.empty(256 * 197, device=dev)
    for _ in range(2000):                                    # clock warm-up with a library kernel
        lib.rovit_layernorm_fwd(p(x), p(xh), p(rs), 256 * 197, 192, 1e-6, sp)
    torch.cuda.synchronize()
    fwd = lambda: lib.rovit_attention_fwd(p(qkv), p(o), p(lse), B, T, H, 64, 0.125, sp)
    bwd = lambda: lib.rovit_attention_bwd(p(qkv), p(o), p(lse), p(dO), p(dqkv), B, T, H, 64, 0.125, sp)
    res = {'batch': B}
    outs = {}
    for r3 in (1, 0):
        lib.rovit_dev_set_knob(KNOB_FWD_R3, r3, 0)
        lib.rovit_dev_set_knob(KNOB_BWD_R3, r3, 0)
        o.fill_(float('nan')); dqkv.fill_(float('nan')); lse.fill_(float('nan'))
        fwd(); bwd()
        torch.cuda.synchronize()
        outs[r3] = (o.clone(), lse.clone(), dqkv.clone())
    res['fwd_bit_identical'] = bool(torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[0][1], outs[1][1]))
    res['bwd_bit_identical'] = bool(torch.equal(outs[0][2].view(torch.int16), outs[1][2].view(torch.int16)))
    if not res['bwd_bit_identical']:
        d = (outs[0][2].float() - outs[1][2].float()).abs()
        res['bwd_max_abs_diff'] = float(d.max()); res['bwd_n_diff'] = int((d > 0).sum())
    for rep in range(3):
        for r3 in (1, 0):
            tag = 'r3' if r3 else 'r4'
            lib.rovit_dev_set_knob(KNOB_FWD_R3, r3, 0)
            lib.rovit_dev_set_knob(KNOB_BWD_R3, r3, 0)
            res.setdefault(f'fwd_us_{tag}', []).append(round(timed(fwd), 2))
            res.setdefault(f'fwd_b2b_us_{tag}', []).append(round(timed(fwd, per_launch=False), 2))
            res.setdefault(f'bwd_us_{tag}', []).append(round(timed(bwd), 2))
            res.setdefault(f'bwd_b2b_us_{tag}', []).append(round(timed(bwd, per_launch=False), 2))
    lib.rovit_dev_set_knob(KNOB_BWD_R3, 0, 0)
    lib.rovit_dev_set_knob(KNOB_SPLIT, 1, 0)             # the passes as separate workgroups (two per item)
    dqkv.fill_(float('nan')); bwd(); torch.cuda.synchronize()
    res['bwd_split_bit_identical'] = bool(torch.equal(dqkv.view(torch.int16), outs[0][2].view(torch.int16)))
    for rep in range(3):
        res.setdefault('bwd_us_r4_split', []).append(round(timed(bwd), 2))
    lib.rovit_dev_set_knob(KNOB_SPLIT, 0, 1)
    for r3 in (1, 0):
        tag = 'r3' if r3 else 'r4'
        lib.rovit_dev_set_knob(KNOB_BWD_R3, r3, 0)
        for bits, name in ((1, 'no_pass1'), (2, 'no_pass2'), (3, 'no_passes')):
            lib.rovit_dev_set_knob(KNOB_DBG, bits, 0)
            res[f'bwd_us_{tag}_{name}'] = round(timed(bwd), 2)
        lib.rovit_dev_set_knob(KNOB_DBG, 0, 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
