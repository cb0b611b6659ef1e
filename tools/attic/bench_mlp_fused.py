"""A/B timing of the MLP half of a block at the benchmark's shape (M = 256 x 197 rows): the two-launch path
(rovit_gemm_nt EPI_GELU + rovit_gemm_resid_ln) against rovit_mlp_fused_fwd, interleaved rounds in one process, device events on the
launch stream.  python tools/bench_mlp_fused.py [M] [rounds]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch  # noqa: E402
from rovit_hip import native  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 197
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device('cuda:0')
    lib = native.load()
    bf = torch.bfloat16
    xhat2 = torch.randn(M, 192, device=dev).to(bf)
    w1 = (torch.randn(768, 192, device=dev) * 0.08).to(bf)
    w2 = (torch.randn(192, 768, device=dev) * 0.05).to(bf)
    b1, b2 = torch.randn(768, device=dev) * 0.3, torch.randn(192, device=dev) * 0.3
    X = torch.randn(M, 192, device=dev)
    act = torch.empty(M, 768, device=dev, dtype=bf)
    dact = torch.empty_like(act)
    xhat = torch.empty(M, 192, device=dev, dtype=bf)
    rstd = torch.empty(M, device=dev)
    ws = torch.empty(lib.rovit_mlp_stream_bytes(), dtype=torch.uint8, device=dev)
    p, sp = native.ptr, native.stream_ptr()
    native.call('rovit_mlp_prepare_stream', p(w1), p(w2), p(ws), sp)

    def two():
        lib.rovit_gemm_nt(p(xhat2), 192, p(w1), 192, M, 768, 192, p(b1), 1, p(act), 768, p(dact), None, 0, None, 0, None, 0, sp)
        lib.rovit_gemm_resid_ln(p(act), 768, p(w2), 768, M, 768, p(b2), p(X), p(xhat), p(rstd), 1e-6, sp)

    def fused(mode):
        a = p(act) if mode >= 1 else None
        d = p(dact) if mode == 2 else None
        return lambda: lib.rovit_mlp_fused_fwd(p(xhat2), p(ws), p(b1), p(b2), a, d, p(X), p(xhat), p(rstd), 1e-6, M, M, sp)

    def with_waves(nw, fn):
        def run():
            lib.rovit_set_mlp_waves(nw)
            fn()
        return run
    variants = {'two_launch': two}
    for nw in (8, 10, 9, 4):
        variants[f'fused_train_w{nw}'] = with_waves(nw, fused(2))
        variants[f'fused_act_only_w{nw}'] = with_waves(nw, fused(1))
        variants[f'fused_inference_w{nw}'] = with_waves(nw, fused(0))
    # backward: fc2 dgrad x gelu' + fc1 dgrad + norm2 backward, two launches against rovit_mlp_fused_bwd
    dY = torch.randn(M, 192, device=dev).to(bf)
    w2t = (torch.randn(768, 192, device=dev) * 0.05).to(bf)
    w1t = (torch.randn(192, 768, device=dev) * 0.05).to(bf)
    dpre = torch.empty(M, 768, device=dev, dtype=bf)
    dX = torch.randn(M, 192, device=dev)
    dXb = torch.empty(M, 192, device=dev, dtype=bf)
    wsb = torch.empty(lib.rovit_mlp_stream_bytes(), dtype=torch.uint8, device=dev)
    native.call('rovit_mlp_prepare_stream', p(w2t), p(w1t), p(wsb), sp)
    dact.uniform_(0, 1)

    def bwd_two():
        lib.rovit_gemm_nt(p(dY), 192, p(w2t), 192, M, 768, 192, None, 3, p(dpre), 768, None, None, 0, p(dact), 768, None, 0, sp)
        lib.rovit_gemm_ln_bwd(p(dpre), 768, p(w1t), 768, M, 768, p(xhat), p(rstd), p(dX), p(dXb), sp)

    def bwd_fused():
        lib.rovit_mlp_fused_bwd(p(dY), p(wsb), p(dact), p(dpre), p(xhat), p(rstd), p(dX), p(dXb), M, sp)
    variants['bwd_two_launch'] = bwd_two
    variants['bwd_fused'] = bwd_fused
    st = torch.cuda.current_stream(dev)

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    a = torch.randn(4096, 4096, device=dev, dtype=bf)
    for _ in range(50):
        a @ a
    res = {k: [] for k in variants}
    for _ in range(rounds):
        for k, fn in variants.items():
            res[k].append(round(timed(fn), 2))
    base = 8.0 * M * 192 + 2.0 * M * 192 + 2.0 * M * 192          # X read + write, xhat out, xhat2 in
    alg = {'two_launch': base + 2.0 * M * 768 * 3}
    for nw in (8, 10, 9, 4):
        alg[f'fused_train_w{nw}'] = base + 2.0 * M * 768 * 2
        alg[f'fused_act_only_w{nw}'] = base + 2.0 * M * 768
        alg[f'fused_inference_w{nw}'] = base
    bbase = 2.0 * M * 192 * 3 + 8.0 * M * 192 + 4.0 * M      # dY, xhat2, dXb; dX read + write; rstd
    alg['bwd_two_launch'] = bbase + 2.0 * M * 768 * 3
    alg['bwd_fused'] = bbase + 2.0 * M * 768 * 2
    out = {k: {'us_min': min(v), 'us_median': sorted(v)[len(v) // 2], 'us_all': v, 'algorithmic_MB': round(alg[k] / 1e6, 1),
               'TBps_at_median': round(alg[k] / (sorted(v)[len(v) // 2] * 1e-6) / 1e12, 2)} for k, v in res.items()}
    print(json.dumps({'M': M, **out}))


if __name__ == '__main__':
    main()
