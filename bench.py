"""Headline benchmark: images/sec of one full RoViT-KAN training step (fwd + joint loss + bwd + gradient all-reduce
+ clip + AdamW) at batch 256 per GPU on synthetic 224x224 images -- BASELINE.json's metric, configs[2]/[3].

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = whole-job images/sec (inputs resident in HBM before the timed region).
`roofline` is measured live (device events on the launch stream) for the dominant kernel class, the bf16 MFMA GEMM
of the attention/MLP linears; `cpu_baseline` times the CPU oracle (a port of the reference's algorithm) on a
bounded sample on this box's host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_FLOP_PER_IMG = 2 * 1253491200            # backbone forward MACs x 2
TRAIN_FLOP_PER_IMG = 7.46e9                  # fwd + bwd (no dgrad into the image)
MFMA_BF16_PEAK_TFLOPS = 2500.0               # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                        # HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def build_optimizer(model, lr=1e-4, wd=1e-4):
    """Same grouping as the reference (training/optimizer.py:7-32): names containing 'backbone' get lr/10."""
    bb = [p for n, p in model.named_parameters() if p.requires_grad and 'backbone' in n]
    hd = [p for n, p in model.named_parameters() if p.requires_grad and 'backbone' not in n]
    return torch.optim.AdamW([{'params': bb, 'lr': lr / 10}, {'params': hd, 'lr': lr}], weight_decay=wd)


def _event_avg_ms(dev, run, iters):
    for _ in range(5):
        run()
    st = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        run()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def wgrad_roofline(dev, iters=30):
    """Dominant kernel of the step: wgrad_kernel<false> (48 launches, ~18 % of the kernel time; rocprofv3 summary in
    profiles/).  Timed here at its largest shape, the fc1 weight gradient G(768,192) = dPre(M,768)^T xhat(M,192) with
    M = 50432 (the fc2 shape moves the same bytes), from device events on the stream it is launched on.  It is
    HBM-bound (32 FLOP per byte read): achieved = algorithmic bytes / duration against the HBM peak."""
    from rovit_hip import native
    lib = native.load()
    M, N, K = 256 * 197, 768, 192
    dY = torch.randn(M, N, device=dev).to(torch.bfloat16)
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    splits = lib.rovit_wgrad_splits(M, N, K)
    slab = torch.empty(lib.rovit_wgrad_workspace_bytes(N, K, splits), dtype=torch.uint8, device=dev)

    def run():
        native.call('rovit_wgrad', native.ptr(dY), N, native.ptr(X), K, M, N, K, splits, 0, native.ptr(slab), native.stream_ptr())
    ms = _event_avg_ms(dev, run, iters)
    flops = 2.0 * M * N * K
    # algorithmic bytes per launch: read dPre (M*N) and xhat (M*K) in bf16, write G (N*K) in fp32
    alg_bytes = 2.0 * (M * N + M * K) + 4.0 * N * K
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': f'wgrad_kernel<false>: fc1 weight gradient, M=50432 N=768 K=192, {splits} M-splits (32 FLOP/B, below the ridge)',
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
            'avg_us': round(ms * 1e3, 2), 'algorithmic_bytes': alg_bytes,
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 2x read correction: profiles/r01_pmc_traffic.txt
            # (includes the fp32 partial-sum slabs the split-M reduction writes: 18.9 MB)
            'traffic': 127.33e6,
            'mfma_tflops': round(flops / (ms * 1e-3) / 1e12, 1)}


def gemm_roofline(dev, iters=30):
    """Second kernel of the step by time (fc1 forward: M=50432, N=768, K=192, GELU + GELU' epilogue; the largest
    single launch), same method."""
    from rovit_hip import native
    M, N, K = 256 * 197, 768, 192
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out2 = torch.empty_like(out)

    def run():
        native.call('rovit_gemm_nt', native.ptr(A), K, native.ptr(W), K, M, N, K, native.ptr(bias), 1, native.ptr(out), N,
                    native.ptr(out2), None, 0, None, 0, None, 0, native.stream_ptr())
    ms = _event_avg_ms(dev, run, iters)
    flops = 2.0 * M * N * K
    # algorithmic bytes per launch: read xhat (M*K) + W (N*K), write act + dact (2*M*N), all bf16
    alg_bytes = 2.0 * (M * K + N * K + 2 * M * N)
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': 'gemm_ws_dma_kernel<GELU>: fc1 forward, M=50432 N=768 K=192 (85 FLOP/B, below the ridge)',
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
            'avg_us': round(ms * 1e3, 2), 'algorithmic_bytes': alg_bytes,
            'traffic': None,
            'mfma_tflops': round(flops / (ms * 1e-3) / 1e12, 1), 'mfma_frac_of_dense_bf16_peak': round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (port of the reference algorithm; the reference itself needs timm and cannot run here) on a
    bounded sample of the same workload: fwd + loss + bwd of the full model, fp32, all host cores."""
    from oracle import ref_cpu
    cores = min(len(os.sched_getaffinity(0)), 32)      # the box's CPU share, not the host's 256 hardware threads
    torch.set_num_threads(cores)
    sd = ref_cpu.init_rovit_state(seed=0)
    params = {k: (v.clone().requires_grad_(True) if 'knots' not in k else v) for k, v in sd.items()}
    B = 16
    x = torch.randn(B, 3, 224, 224)
    y = torch.randint(0, 4, (B,))

    def step():
        out = ref_cpu.rovit_forward(x, params, 4)
        ref_cpu.joint_loss(out, y, y, 4)['total_loss'].backward()
    step()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds_budget and n < 20:
        step()
        n += 1
    dt = time.time() - t0
    return {'value': round(n * B / dt, 2), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n} steps of batch {B}, fwd+loss+bwd, fp32, vectorised KAN restatement (oracle/ref_cpu.py)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--buckets', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--roofline-only', action='store_true',
                    help='run only the two roofline kernel measurements (the command profiles/r01_roofline_kernel_stats.csv is taken from)')
    args = ap.parse_args()
    if args.roofline_only:
        dev = torch.device('cuda:0')
        print(json.dumps({'roofline': wgrad_roofline(dev), 'roofline_gemm': gemm_roofline(dev)}), flush=True)
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force_dist = os.environ.get('ROVIT_FORCE_DIST') == '1'      # exercise the RCCL code path on a single GPU
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    from models.rovit_kan import RoViTKAN
    from rovit_hip.losses import JointLoss
    from rovit_hip.parallel import GradSync
    from rovit_hip.optim import RoViTAdamW

    torch.manual_seed(0)                                 # identical replica on every rank
    model = RoViTKAN(pretrained=False).to(dev).train()   # random-init DeiT-Tiny + KAN head, dropout 0.3, stage 4
    model.curriculum_stage = 4
    opt = RoViTAdamW(model, lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0)   # clip_grad_norm_(1.0) + AdamW, backbone at lr/10
    loss_fn = JointLoss(1.0, 0.5, 0.5, 2.0, torch.ones(4, device=dev))
    sync = GradSync(model, buckets=args.buckets, force=force_dist)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    images = torch.randn(args.batch, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 4, (args.batch,), device=dev, generator=g)

    def step():
        out = model(images)
        loss = loss_fn(out, labels, labels, 4)['total_loss']
        opt.zero_grad(set_to_none=True)
        loss.backward()
        sync.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)

    if rank == 0:
        ips = world * args.batch * args.steps / dt
        res = {
            'metric': 'images/sec fwd+bwd, DeiT-Tiny+KAN 224^2, batch 256 per GPU',
            'value': round(ips, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'full RoViT-KAN (DeiT-Tiny + KAN + 3 heads) stage 4, train mode, batch %d/GPU, '
                                   '224x224x3 randn images, random-init weights' % args.batch,
                       'step': 'fwd + JointLoss + bwd + grad all-reduce + clip_grad_norm(1.0) + AdamW',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world}', 'grad_buckets': args.buckets,
                       'backbone_mfma_frac_of_step': round(ips / world * TRAIN_FLOP_PER_IMG / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
            'final_loss': round(final_loss, 5),
        }
        res['roofline'] = wgrad_roofline(dev)
        res['roofline_gemm'] = gemm_roofline(dev)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
