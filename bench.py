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
FP32_PEAK_TFLOPS = 157.3                     # fp32 vector = fp32 matrix peak (MI355X_MICROARCH.md)


def build_optimizer(model, lr=1e-4, wd=1e-4):
    """Same grouping as the reference (training/optimizer.py:7-32): names containing 'backbone' get lr/10."""
    bb = [p for n, p in model.named_parameters() if p.requires_grad and 'backbone' in n]
    hd = [p for n, p in model.named_parameters() if p.requires_grad and 'backbone' not in n]
    return torch.optim.AdamW([{'params': bb, 'lr': lr / 10}, {'params': hd, 'lr': lr}], weight_decay=wd)


def _warm_clocks(dev, seconds=0.3):
    """Keep the device busy with a kernel of THIS library that none of the roofline sections measures (the plain LayerNorm
    forward: one launch per step in the model, 77 MB here) until the clocks have left the idle state: the first ~100 ms after
    an idle period run 10-15 % slower (measured: the same launch 109 us cold against 95 us after the training loop).
    (Round 2 warmed up with a rocBLAS GEMM, which then dominated the rocprof summaries of this command.)"""
    from rovit_hip import native
    rows = 256 * 197
    x = torch.randn(rows, 192, device=dev)
    xh = torch.empty(rows, 192, device=dev, dtype=torch.bfloat16)
    rs = torch.empty(rows, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            native.call('rovit_layernorm_fwd', native.ptr(x), native.ptr(xh), native.ptr(rs), rows, 192, 1e-6, native.stream_ptr())
        torch.cuda.synchronize(dev)


def _event_avg_ms(dev, run, iters, per_launch=True):
    """Average duration of one call of `run` from device events on the stream it launches on.  per_launch=True brackets
    EVERY launch with its own pair of events (a kernel's whole life, first workgroup in to last workgroup out -- what
    rocprofv3 --kernel-trace reports per dispatch); per_launch=False times the back-to-back train, where one launch's
    ramp hides in the previous launch's tail (throughput view; 10-18 % shorter for these 20-100 us kernels)."""
    for _ in range(5):
        run()
    st = torch.cuda.current_stream(dev)
    if not per_launch:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            run()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(st)
        run()
        b.record(st)
    evs[-1][1].synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters


def _pmc_traffic(kernel_key):
    """HBM bytes per launch measured with rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 2x read
    correction) for `python bench.py --roofline-only` (tools/pmc_collect.py): read from the tracked
    profiles/r03_pmc_traffic.json (round 2's file as a fallback), or None."""
    for name in ('r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                t = json.load(f).get(kernel_key, {}).get('traffic_bytes')
            if t is not None:
                return t
        except (OSError, ValueError):
            pass
    return None


def _entry(dev, run, alg_bytes, flops, kernel, pmc_key, iters=30, **extra):
    """One roofline entry: `run` launches the kernel once on the current stream; duration from device events around every
    launch; HBM fraction on ALGORITHMIC bytes, matrix-core fraction on algorithmic FLOPs, PMC traffic from profiles/."""
    ms = _event_avg_ms(dev, run, iters)
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    e = {'bound': 'hbm', 'kernel': kernel, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
         'frac': round(gbs / HBM_PEAK_GBS, 4), 'avg_us': round(ms * 1e3, 2), 'algorithmic_bytes': float(alg_bytes),
         'traffic': _pmc_traffic(pmc_key)}
    if flops:
        tf = flops / (ms * 1e-3) / 1e12
        e['mfma_tflops'] = round(tf, 1)
        e['mfma_frac_of_dense_bf16_peak'] = round(tf / MFMA_BF16_PEAK_TFLOPS, 4)
    e.update(extra)
    return e


def attn_roofline(dev):
    """north_star: "images/sec ... as fraction of the attention-GEMM roofline".  One entry per kernel of the attention half of
    a block at the benchmark's shape (256 images x 197 tokens, 3 heads x 64): QKV projection, softmax(QK^T)V forward and
    backward, output projection (+ residual + norm2), and the two dgrads.  Every one of them sits below the chip's
    ~310 FLOP/B ridge (K = 192 / 576, head dim 64), so each is priced against the HBM peak on its algorithmic bytes; the
    achieved matrix-core rate is reported beside it (mfma_frac_of_dense_bf16_peak) on algorithmic (197-token) FLOPs."""
    from rovit_hip import native
    p, sp = native.ptr, native.stream_ptr()
    lib = native.load()
    B, T, H = 256, 197, 3
    M = B * T
    bf = torch.bfloat16
    xhat = torch.randn(M, 192, device=dev).to(bf)
    wqkv = (torch.randn(576, 192, device=dev) * 0.05).to(bf)
    bqkv = torch.randn(576, device=dev) * 0.1
    qkv = torch.empty(M, 576, device=dev, dtype=bf)
    o = torch.empty(M, 192, device=dev, dtype=bf)
    lse = torch.empty(B, H, T, device=dev)
    wproj = (torch.randn(192, 192, device=dev) * 0.05).to(bf)
    bproj = torch.randn(192, device=dev) * 0.1
    X = torch.randn(M, 192, device=dev)
    xhat2 = torch.empty(M, 192, device=dev, dtype=bf)
    rstd = torch.empty(M, device=dev)
    dO = torch.randn(M, 192, device=dev).to(bf)
    dqkv = torch.empty(M, 576, device=dev, dtype=bf)
    dX = torch.randn(M, 192, device=dev)
    dXb = torch.empty(M, 192, device=dev, dtype=bf)
    wqkvT = (torch.randn(192, 576, device=dev) * 0.05).to(bf)
    res = {}
    res['qkv_fwd'] = _entry(dev, lambda: lib.rovit_gemm_nt(p(xhat), 192, p(wqkv), 192, M, 576, 192, p(bqkv), 0, p(qkv), 576, None, None, 0, None, 0, None, 0, sp),
                            2.0 * M * (192 + 576) + 2.0 * 576 * 192, 2.0 * M * 576 * 192,
                            'gemm_ws_dma_kernel<0>: QKV projection, M=50432 N=576 K=192', 'qkv_fwd')
    res['attention_fwd'] = _entry(dev, lambda: lib.rovit_attention_fwd(p(qkv), p(o), p(lse), B, T, H, 64, 0.125, sp),
                                  2.0 * M * 576 + 2.0 * M * 192 + 4.0 * B * H * T, 4.0 * B * H * T * T * 64,
                                  'attn_fwd_kernel: softmax(QK^T/8)V, one workgroup per (image, head), 197x197 tile on chip', 'attention_fwd')
    res['proj_fwd_resid_ln'] = _entry(dev, lambda: lib.rovit_gemm_resid_ln(p(o), 192, p(wproj), 192, M, 192, p(bproj), p(X), p(xhat2), p(rstd), 1e-6, sp),
                                      2.0 * M * 192 * 2 + 8.0 * M * 192 + 4.0 * M + 2.0 * 192 * 192, 2.0 * M * 192 * 192,
                                      'gemm_ws_kernel<6,1,64,5>: output projection + residual add + norm2, M=50432 N=K=192', 'proj_fwd_resid_ln')
    res['proj_dgrad'] = _entry(dev, lambda: lib.rovit_gemm_nt(p(xhat), 192, p(wproj), 192, M, 192, 192, None, 0, p(o), 192, None, None, 0, None, 0, None, 0, sp),
                               2.0 * M * 192 * 2 + 2.0 * 192 * 192, 2.0 * M * 192 * 192,
                               'gemm_ws_dma_kernel<0>: output-projection dgrad, M=50432 N=K=192', 'proj_dgrad')
    lib.rovit_attention_fwd(p(qkv), p(o), p(lse), B, T, H, 64, 0.125, sp)
    res['attention_bwd'] = _entry(dev, lambda: lib.rovit_attention_bwd(p(qkv), p(o), p(lse), p(dO), p(dqkv), B, T, H, 64, 0.125, sp),
                                  2.0 * M * 576 * 2 + 2.0 * M * 192 * 2 + 4.0 * B * H * T, 10.0 * B * H * T * T * 64,
                                  'attn_bwd_kernel: dQ, dK, dV with the probabilities recomputed on chip (algorithmic FLOPs: 5 products)', 'attention_bwd')
    dXin = dX.to(bf)
    # round 4: the residual gradient enters and leaves as bf16 rows (the step's form): dqkv + xhat + incoming gradient read, bf16 rows written
    res['qkv_dgrad_ln_bwd'] = _entry(dev, lambda: lib.rovit_gemm_ln_bwd(p(dqkv), 576, p(wqkvT), 576, M, 576, p(xhat), p(rstd), None, p(dXin), p(dXb), sp),
                                     2.0 * M * 576 + 2.0 * M * 192 * 3 + 4.0 * M + 2.0 * 192 * 576, 2.0 * M * 576 * 192,
                                     'gemm_kdma_kernel<18,6>: QKV dgrad + norm1 backward, bf16 residual gradient in and out, M=50432 N=192 K=576', 'qkv_dgrad_ln_bwd')
    return res


def mlp_roofline(dev):
    """The MLP half of a block as ONE launch each way (round 3, csrc/mlp_fused.hip): fc1 + GELU + fc2 + residual + next LayerNorm
    forward (training: act and gelu' written once, never re-read) and fc2 dgrad x gelu' + fc1 dgrad + norm2 backward."""
    from rovit_hip import native
    p, sp = native.ptr, native.stream_ptr()
    lib = native.load()
    M = 256 * 197
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
    wsb = torch.empty_like(ws)
    native.call('rovit_mlp_prepare_stream', p(w1), p(w2), p(ws), sp)
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
    native.call('rovit_mlp_prepare_stream', p(w2t), p(w1t), p(wsb), sp)
    dY = torch.randn(M, 192, device=dev).to(bf)
    dpre = torch.empty(M, 768, device=dev, dtype=bf)
    dX = torch.randn(M, 192, device=dev)
    dXb = torch.empty(M, 192, device=dev, dtype=bf)
    wbytes = 2.0 * 2 * 768 * 192
    res = {}
    res['fused_fwd_train'] = _entry(dev, lambda: lib.rovit_mlp_fused_fwd(p(xhat2), p(ws), p(b1), p(b2), p(act), p(dact), p(X), p(xhat), p(rstd), 1e-6, M, M, sp),
                                    2.0 * M * 192 * 2 + 2.0 * M * 768 * 2 + 8.0 * M * 192 + 4.0 * M + wbytes, 4.0 * M * 768 * 192,
                                    'mlp_fused_kernel<0,2,8,false,true> (in-wave pipeline, GELU table): fc1 + GELU + fc2 + residual + LayerNorm, act and gelu\' kept, M=50432', 'mlp_fused_fwd_train')
    res['fused_fwd_inference'] = _entry(dev, lambda: lib.rovit_mlp_fused_fwd(p(xhat2), p(ws), p(b1), p(b2), None, None, p(X), p(xhat), p(rstd), 1e-6, M, M, sp),
                                        2.0 * M * 192 * 2 + 8.0 * M * 192 + 4.0 * M + wbytes, 4.0 * M * 768 * 192,
                                        'mlp_fused_kernel<0,0,8,false,true>: the same, nothing kept (inference)', 'mlp_fused_fwd_inference')
    # the step's forward kernel since late round 3: everything of a block behind the attention + the next block's qkv projection
    o = torch.randn(M, 192, device=dev).to(bf)
    wp = (torch.randn(192, 192, device=dev) * 0.07).to(bf)
    wq = (torch.randn(576, 192, device=dev) * 0.07).to(bf)
    bp_, bq = torch.randn(192, device=dev) * 0.2, torch.randn(576, device=dev) * 0.2
    wst = torch.empty_like(ws)
    native.call('rovit_mlp_prepare_stream_tail', p(w1), p(w2), p(wp), p(wq), p(wst), sp)
    xh2, r2 = torch.empty(M, 192, device=dev, dtype=bf), torch.empty(M, device=dev)
    qkv = torch.empty(M, 576, device=dev, dtype=bf)
    tail_w = wbytes + 2.0 * 192 * 192 + 2.0 * 576 * 192
    tail_flops = 4.0 * M * 768 * 192 + 2.0 * M * 192 * 192 + 2.0 * M * 192 * 576
    res['block_tail_train'] = _entry(
        dev, lambda: lib.rovit_block_tail_fwd(p(o), p(wst), p(bp_), p(b1), p(b2), p(X), p(xh2), p(r2), p(act), p(dact), p(xhat), p(rstd), p(bq), p(qkv),
                                              1e-6, M, M, sp),
        2.0 * M * 192 * 3 + 8.0 * M * 192 + 2.0 * M * 768 * 2 + 2.0 * M * 576 + 8.0 * M + tail_w, tail_flops,
        'mlp_fused_kernel<0,2,8,false,true,true> (block tail): proj + residual + norm2 + fc1 + GELU + fc2 + residual + next norm1 + next qkv, M=50432',
        'block_tail_train')
    res['block_tail_inference'] = _entry(
        dev, lambda: lib.rovit_block_tail_fwd(p(o), p(wst), p(bp_), p(b1), p(b2), p(X), None, None, None, None, p(xhat), p(rstd), p(bq), p(qkv),
                                              1e-6, M, M, sp),
        2.0 * M * 192 * 2 + 8.0 * M * 192 + 2.0 * M * 576 + 4.0 * M + tail_w, tail_flops,
        'mlp_fused_kernel<0,0,8,false,true,true>: the same, nothing kept (inference)', 'block_tail_inference')
    dact.uniform_(0, 1)
    # round 4: dX = NULL -- the residual gradient travels in bf16 (dY is the incoming gradient; read once more by the row pass: L2)
    res['fused_bwd'] = _entry(dev, lambda: lib.rovit_mlp_fused_bwd(p(dY), p(wsb), p(dact), p(dpre), p(xhat2), p(rstd), None, p(dXb), M, sp),
                              2.0 * M * 192 * 3 + 2.0 * M * 768 * 2 + 4.0 * M + wbytes, 4.0 * M * 768 * 192,
                              'mlp_fused_kernel<1,1,8>: fc2 dgrad x gelu\' + fc1 dgrad + norm2 backward, dpre written once, bf16 residual gradient, M=50432', 'mlp_fused_bwd')
    return res


def wgrad_roofline(dev, iters=30):
    """Dominant kernel of the step (rocprofv3 summaries in profiles/): wgrad_kernel<192,192,..,4>, the ONE weight-gradient
    launch per transformer block: G = dY^T A for the qkv, fc2, fc1 and proj linears together (M = 50432 rows, 12 output
    tiles of 192 x 192 per M-split, eight waves per workgroup, 16 splits = 192 workgroups: the step's configuration).  Timed from device events on the stream it is launched on.  HBM-bound
    (<= 48 FLOP per byte read): achieved = algorithmic bytes / duration against the HBM peak."""
    import ctypes as C
    from rovit_hip import native
    lib = native.load()
    M, splits = 256 * 197, 16
    shapes = [(576, 192), (192, 768), (768, 192), (192, 192)]       # (N, K) of qkv, fc2, fc1, proj
    dY = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n, _ in shapes]
    A = [torch.randn(M, k, device=dev).to(torch.bfloat16) for _, k in shapes]
    ws = [torch.empty(lib.rovit_wgrad_workspace_bytes(n, k, splits), dtype=torch.uint8, device=dev) for n, k in shapes]
    arr = lambda xs: (C.c_int * len(xs))(*xs)
    a_dy, a_a, a_ws = native.ptr_array(dY), native.ptr_array(A), native.ptr_array(ws)
    ldy, lda, Ns, Ks = arr([n for n, _ in shapes]), arr([k for _, k in shapes]), arr([n for n, _ in shapes]), arr([k for _, k in shapes])
    # the step's operand layouts: act (A of fc2) and dpre (dY of fc1) are chunk-major, written so by the one-launch MLP kernels
    a_blk, y_blk = arr([0, 1, 0, 0]), arr([0, 0, 1, 0])

    def run():
        native.call('rovit_wgrad_multi_ex', a_dy, ldy, a_a, lda, Ns, Ks, a_ws, a_blk, y_blk, 4, M, splits, native.stream_ptr())
    _warm_clocks(dev)
    ms = _event_avg_ms(dev, run, iters)
    ms_train = _event_avg_ms(dev, run, iters, per_launch=False)
    flops = sum(2.0 * M * n * k for n, k in shapes)
    # algorithmic bytes per launch: read dY (M*N) and A (M*K) in bf16 once, write one fp32 G (N*K), per
    # problem (the partial slabs the kernel really writes count as traffic, not as algorithmic bytes)
    alg_bytes = sum(2.0 * M * (n + k) + 4.0 * n * k for n, k in shapes)
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': f'wgrad_kernel<192,192,false,false,4>: weight gradients of one block (qkv+fc2+fc1+proj) in one launch, M=50432, {splits} M-splits',
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
            'avg_us': round(ms * 1e3, 2), 'avg_us_back_to_back': round(ms_train * 1e3, 2), 'algorithmic_bytes': alg_bytes,
            'traffic': _pmc_traffic('wgrad_kernel<192,192>'),
            'mfma_tflops': round(flops / (ms * 1e-3) / 1e12, 1)}


def backbone_only(dev, batch=256, iters=10):
    """BASELINE.json configs[1]: the DeiT-Tiny backbone alone at batch 256, bf16 -- inference forward and forward + backward with
    the surrogate loss features.float().square().mean(), device events around whole passes."""
    from models.backbone import DeiTTinyBackbone
    torch.manual_seed(0)
    bb = DeiTTinyBackbone(pretrained=False).to(dev)
    x = torch.randn(batch, 3, 224, 224, device=dev)
    bb.eval()
    with torch.no_grad():
        ms_f = _event_avg_ms(dev, lambda: bb(x), iters, per_launch=False)
    bb.train()

    def fb():
        for p in bb.parameters():
            p.grad = None
        bb(x).float().square().mean().backward()
    ms_fb = _event_avg_ms(dev, fb, iters, per_launch=False)
    return {'batch': batch, 'fwd_inference_ms': round(ms_f, 3), 'fwd_inference_images_per_sec': round(batch / (ms_f * 1e-3), 1),
            'fwd_bwd_ms': round(ms_fb, 3), 'fwd_bwd_images_per_sec': round(batch / (ms_fb * 1e-3), 1),
            'fwd_mfma_frac_of_dense_bf16_peak': round(FWD_FLOP_PER_IMG * batch / (ms_f * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            'fwd_bwd_mfma_frac_of_dense_bf16_peak': round(TRAIN_FLOP_PER_IMG * batch / (ms_fb * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}


def patch_roofline(dev):
    """The PatchEmbed pair (timm PatchEmbed = conv k16 s16, reached through models/backbone.py:12-25) at batch 256: the forward GEMM
    gathers its fp32 pixels from the NCHW images (154 MB read, 38.5 MB of token rows written), the weight gradient re-reads them."""
    from rovit_hip import native
    p, sp = native.ptr, native.stream_ptr()
    lib = native.load()
    B, T = 256, 197
    img = torch.randn(B, 3, 224, 224, device=dev)
    W = (torch.randn(192, 768, device=dev) * 0.03).to(torch.bfloat16)
    bias, pos = torch.randn(192, device=dev) * 0.1, torch.randn(T, 192, device=dev) * 0.02
    X = torch.empty(B * T, 192, device=dev)
    dXb = torch.randn(B * T, 192, device=dev).to(torch.bfloat16)
    M = B * (T - 1)
    splits = lib.rovit_wgrad_splits(M, 192, 768)
    ws = torch.empty(lib.rovit_wgrad_workspace_bytes(192, 768, splits), dtype=torch.uint8, device=dev)
    res = {}
    res['patch_embed_fwd'] = _entry(dev, lambda: lib.rovit_patch_embed_fwd(p(img), p(W), p(bias), p(pos), p(X), B, T, sp),
                                    4.0 * B * 3 * 224 * 224 + 4.0 * M * 192 + 2.0 * 192 * 768, 2.0 * M * 192 * 768,
                                    'gemm_ws_kernel<12,2,32,7>: PatchEmbed as a GEMM whose pixels are gathered from the fp32 images, M=50176 N=192 K=768', 'patch_embed_fwd')
    res['patch_embed_wgrad'] = _entry(dev, lambda: lib.rovit_patch_embed_wgrad(p(dXb), 192, p(img), B, T, 192, splits, p(ws), sp),
                                      4.0 * B * 3 * 224 * 224 + 2.0 * M * 192 + 4.0 * 192 * 768, 2.0 * M * 192 * 768,
                                      'wgrad_kernel<192,96,true,true,2>: PatchEmbed weight gradient, pixels gathered again from the fp32 images', 'patch_embed_wgrad')
    return res


def fp32_mode(dev, batch=256, iters=5):
    """The reference-precision forward (DeiTTiny.precision = 'fp32': every product and sum in fp32, GEMMs and attention on
    v_mfma_f32_32x32x2_f32) at the benchmark's batch: the only mode that meets north_star's 1e-3 / identical-argmax clause
    end to end (tests/test_gpu_round2.py::test_fp32_reference_precision_mode...).  Whole model forward, device events."""
    from models.rovit_kan import RoViTKAN
    torch.manual_seed(0)
    m = RoViTKAN(pretrained=False).to(dev).eval()
    m.curriculum_stage = 4
    m.backbone.model.precision = 'fp32'
    x = torch.randn(batch, 3, 224, 224, device=dev)
    with torch.no_grad():
        ms = _event_avg_ms(dev, lambda: m(x), iters, per_launch=False)
    tf = FWD_FLOP_PER_IMG * batch / (ms * 1e-3) / 1e12
    return {'ms_per_forward': round(ms, 3), 'images_per_sec': round(batch / (ms * 1e-3), 1), 'batch': batch,
            'fp32_tflops': round(tf, 1), 'frac_of_fp32_matrix_peak': round(tf / FP32_PEAK_TFLOPS, 4),
            'kernels': 'gemm_f32_mfma_kernel (128x64 tiles, five workgroups per CU, LayerNorms folded into the neighbouring GEMMs), attn_f32_mfma_kernel (half an (image, head) per workgroup, two per CU): v_mfma_f32_32x32x2_f32, exact fp32 FMA chains; two half-batch chains on two streams from batch 192'}


def kan_roofline(dev, iters=30):
    """KAN spline head (north_star: HBM roofline, no MFMA), kernels only (direct C-ABI calls):
    C5 = BASELINE.json configs[4] (num_knots 32, batch 512: three per-layer launches, the faster path at that size) and
    the streaming shapes batch 65536 (one launch on the matrix cores, rovit_kan_stack_fwd_mfma).  Algorithmic bytes: the
    weights once per launch + x and every layer output once."""
    import ctypes as C
    from models.kan import KANSeverityModule
    from rovit_hip import native
    from rovit_hip.functions import ACT_RELU, ACT_SIGMOID3
    from rovit_hip.native import ptr, ptr_array
    lib = native.load()
    layers = [192, 64, 16, 1]
    res = {}
    _warm_clocks(dev)
    for key, G, B in (('c5_g32_b512', 32, 512), ('stream_g5_b65536', 5, 65536), ('stream_g32_b65536', 32, 65536), ('c3_g5_b256', 5, 256)):
        m = KANSeverityModule(layers, G, 3).to(dev)
        nb, n = G + 2, 3
        x = torch.randn(B, 192, device=dev)
        outs = [torch.empty(B, layers[l + 1], device=dev) for l in range(n)]
        arr = lambda xs: (C.c_int * len(xs))(*xs)
        sp = native.stream_ptr()
        fused = B >= m.fused_min_batch
        mfma = fused and B >= m.mfma_min_batch and all(p[2] is not None for p in m._prepared())
        if mfma:
            prep = m._prepared()
            margs = (ptr(x), ptr_array([p[2] for p in prep]), ptr_array([l.knots for l in m.kan_layers]),
                     ptr_array([l.linear.bias for l in m.kan_layers]), ptr_array(outs), B, arr(layers),
                     arr([l.knots.numel() for l in m.kan_layers]), arr([ACT_RELU, ACT_RELU, ACT_SIGMOID3]), n, sp)
            run = lambda: lib.rovit_kan_stack_fwd_mfma(*margs)
        elif fused:
            prep = m._prepared()
            args = (ptr(x), ptr_array([p[0] for p in prep]), ptr_array([l.knots for l in m.kan_layers]), ptr_array([p[1] for p in prep]),
                    ptr_array([l.linear.bias for l in m.kan_layers]), ptr_array(outs), B, arr(layers),
                    arr([l.knots.numel() for l in m.kan_layers]), arr([ACT_RELU, ACT_RELU, ACT_SIGMOID3]), n, sp)
            run = lambda: lib.rovit_kan_stack_fwd(*args)
        else:
            ins = [x] + outs[:-1]
            raw = [(ptr(ins[i]), ptr(l.spline_weights), ptr(l.knots), ptr(l.linear.weight), ptr(l.linear.bias), ptr(outs[i]), B, l.in_features,
                    l.out_features, l.knots.numel(), ACT_SIGMOID3 if i == n - 1 else ACT_RELU, sp) for i, l in enumerate(m.kan_layers)]

            def run():
                for r in raw:
                    lib.rovit_kan_layer_fwd(*r)
        ms = _event_avg_ms(dev, run, iters)
        w_bytes = sum(a * b * nb + a * b + b for a, b in zip(layers[:-1], layers[1:])) * 4
        act_bytes = B * (layers[0] + (sum(layers[1:]) if fused else 2 * sum(layers[1:-1]) + layers[-1])) * 4
        alg = float(w_bytes + act_bytes)
        gbs = alg / (ms * 1e-3) / 1e9
        kname = ('kan_stack_mfma_kernel (one launch, fp32 MFMA)' if mfma else 'kan_stack_fwd_kernel (one launch)' if fused
                 else 'kan_fwd_kernel x3 (per layer)')
        res[key] = {'bound': 'hbm', 'kernel': kname,
                    'num_knots': G, 'batch': B, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / HBM_PEAK_GBS, 5), 'avg_us': round(ms * 1e3, 2), 'algorithmic_bytes': alg,
                    'traffic': _pmc_traffic(f'kan_stack_mfma_kernel<{4 if G == 5 else 18}>') if mfma else
                    (_pmc_traffic('kan_fwd_' + key) if not fused else None)}
        # the honest compute roofline of this kernel class (121 FLOP/B against a 20 FLOP/B fp32 ridge): fp32
        # FLOPs on the USEFUL terms -- per (sample, input, output) the 4 live basis products + the Linear term -- against the
        # 157.3 TFLOP/s fp32 peak (vector = matrix rate on gfx950)
        useful = 2.0 * B * sum(a * b * 5 for a, b in zip(layers[:-1], layers[1:]))
        res[key]['useful_fp32_tflops'] = round(useful / (ms * 1e-3) / 1e12, 2)
        res[key]['useful_frac_of_fp32_peak'] = round(useful / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
        if mfma:   # dense matrix-core work (structural zeros included) against the fp32 MFMA peak, for the record
            S = 8 if G == 5 else 36
            mf = 2.0 * B * sum(a * S * 32 * (2 if b > 32 else 1) for a, b in zip(layers[:-1], layers[1:]))
            res[key]['mfma_f32_tflops'] = round(mf / (ms * 1e-3) / 1e12, 1)
            res[key]['mfma_frac_of_f32_matrix_peak'] = round(mf / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
    res['c5_g32_b512_fwd_bwd'] = kan_fwd_bwd_roofline(dev, 32, 512, iters)
    res['c3_g5_b256_fwd_bwd'] = kan_fwd_bwd_roofline(dev, 5, 256, iters)
    res['head_phase_b256'] = head_phase_roofline(dev, 256, iters)
    res['head_phase_c5_g32_b512'] = head_phase_roofline(dev, 512, iters, num_knots=32)
    return res


def head_phase_roofline(dev, B, iters=30, num_knots=5):
    """What the training step runs at BASELINE.json configs[2] since round 4: the three heads AND the KAN stack as one forward launch and
    a two-launch backward (csrc/head_phase.hip), kernels only (direct C-ABI calls on preallocated buffers, dropout drawn in the kernel).
    Algorithmic bytes: every parameter once per launch (forward, per-sample backward) / read once and its gradient written once
    (parameter-gradient launch), features, hidden activations, outputs and their gradients once."""
    import ctypes as C
    from models.rovit_kan import RoViTKAN
    from rovit_hip import native
    from rovit_hip.functions import HeadPhaseFn
    lib = native.load()
    m = RoViTKAN(pretrained=False, kan_num_knots=num_knots).to(dev).train()
    k = m.kan_module
    nl, hid = len(k.kan_layers), 128
    cfg = {'stage': 4, 'masks': None, 'drop_p': 0.3, 'seed': 1, 'offset': 0, 'kan_dims': list(k.layers_dims),
           'kan_knots': [l.knots for l in k.kan_layers], 'kan_acts': [2 if i == nl - 1 else 1 for i in range(nl)], 'grad_views': None}
    hp, kp = [p.detach() for p in m._head_params()], [p.detach() for p in m._kan_params()]
    feats = torch.randn(B, 192, device=dev)
    d = HeadPhaseFn._desc(feats, cfg, hp, kp)
    bufs = {n: torch.empty(*s, device=dev) for n, s in (('hidden', (3, B, hid)), ('cls', (B, 4)), ('ord', (B, 3)), ('mu', (B, 1)), ('lv', (B, 1)))}
    kouts = [torch.empty(B, w, device=dev) for w in k.layers_dims[1:]]
    d.hidden, d.cls, d.ord, d.mu, d.lv = (bufs[n].data_ptr() for n in ('hidden', 'cls', 'ord', 'mu', 'lv'))
    for l in range(nl):
        d.kan_out[l] = kouts[l].data_ptr()
    g = [torch.randn_like(t) for t in (bufs['cls'], bufs['ord'], bufs['mu'], bufs['lv'], kouts[-1])]
    d.g_cls, d.g_ord, d.g_mu, d.g_lv, d.g_kan = (t.data_ptr() for t in g)
    dfeat = torch.empty(B, 192, device=dev)
    scratch = torch.empty(3 * B * hid + B * sum(k.layers_dims[1:]), device=dev)
    d.d_features, d.dpre = dfeat.data_ptr(), scratch.data_ptr()
    off = 3 * B * hid
    for l in range(nl):
        d.kan_gz[l] = scratch.data_ptr() + 4 * off
        off += B * k.layers_dims[l + 1]
    grads = [torch.empty_like(p) for p in hp + kp]
    for i in range(14):
        d.head_grads[i] = grads[i].data_ptr()
    for l in range(nl):
        d.kan_dw[l], d.kan_dlw[l], d.kan_dlb[l] = (grads[14 + 3 * l + q].data_ptr() for q in range(3))
    sp = native.stream_ptr()
    _warm_clocks(dev)
    t_fwd = _event_avg_ms(dev, lambda: lib.rovit_head_phase_fwd(C.byref(d), sp), iters)
    lib.rovit_head_phase_fwd(C.byref(d), sp)
    d.want_param_grads = 0
    t_dx = _event_avg_ms(dev, lambda: lib.rovit_head_phase_bwd(C.byref(d), sp), iters)
    t_dw = _event_avg_ms(dev, lambda: lib.rovit_head_phase_bwd_params(C.byref(d), sp), iters)
    p_bytes = 4 * sum(p.numel() for p in hp + kp)
    io = 4 * B * (192 + 3 * hid + 4 + 3 + 1 + 1 + sum(k.layers_dims[1:]))
    alg = {'fwd': p_bytes + io, 'bwd_per_sample': p_bytes + 2 * io, 'bwd_params': 2 * p_bytes + 2 * io}
    out = {'bound': 'hbm', 'kernel': 'head_phase_fwd_kernel / head_phase_bwd_dx_kernel / head_phase_dw_kernel: the three heads + the KAN stack '
           '(192-64-16-1, num_knots %d), one workgroup per sample' % num_knots, 'batch': B, 'num_knots': num_knots, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'traffic': None}
    for key, t in (('fwd', t_fwd), ('bwd_per_sample', t_dx), ('bwd_params', t_dw)):
        out[key + '_us'] = round(t * 1e3, 2)
        out[key + '_traffic'] = _pmc_traffic('head_phase_' + key) if num_knots == 5 and B == 256 else None
        out[key + '_algorithmic_bytes'] = float(alg[key])
        out[key + '_frac'] = round(alg[key] / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    out['avg_us'] = round((t_fwd + t_dx + t_dw) * 1e3, 2)
    out['achieved'] = round(sum(alg.values()) / ((t_fwd + t_dx + t_dw) * 1e-3) / 1e9, 1)
    out['frac'] = round(out['achieved'] / HBM_PEAK_GBS, 5)
    out['algorithmic_bytes'] = float(sum(alg.values()))
    out['note'] = ('launch- and latency-bound at the reference sizes (0.74 MB of parameters, 256 samples): each workgroup streams the parameters '
                   'from L2 once, ~5 us at a CU\'s L2 port; before round 4 the same work was 7 + 7 launches, 64 + 118 us in the step')
    return out


def kan_fwd_bwd_roofline(dev, G, B, iters=30):
    """BASELINE.json configs[4] / [2] as the training step runs them: KANSeverityModule forward AND backward, kernels only
    (direct C-ABI calls on preallocated buffers, the launches KANStackFn / KANLayerFn make: the module call itself is
    host-bound at these sizes).  Algorithmic bytes: forward = weights + x + every layer output once;
    backward = read weights, write their gradients, read the activations and output gradients, write the input gradients."""
    from models.kan import KANSeverityModule
    from rovit_hip import native
    from rovit_hip.functions import ACT_RELU, ACT_SIGMOID3
    from rovit_hip.native import ptr
    lib = native.load()
    layers = [192, 64, 16, 1]
    nb, n = G + 2, 3
    m = KANSeverityModule(layers, G, 3).to(dev)
    x = torch.randn(B, 192, device=dev)
    outs = [torch.empty(B, layers[l + 1], device=dev) for l in range(n)]
    gout = torch.randn(B, 1, device=dev)
    dxs = [torch.empty(B, layers[l], device=dev) for l in range(n)]
    dws = [torch.empty_like(l.spline_weights) for l in m.kan_layers]
    dlw = [torch.empty_like(l.linear.weight) for l in m.kan_layers]
    dlb = [torch.empty_like(l.linear.bias) for l in m.kan_layers]
    sp = native.stream_ptr()
    ins = [x] + outs[:-1]
    acts = [ACT_RELU, ACT_RELU, ACT_SIGMOID3]
    fwd = [(ptr(ins[i]), ptr(l.spline_weights), ptr(l.knots), ptr(l.linear.weight), ptr(l.linear.bias), ptr(outs[i]), B, l.in_features,
            l.out_features, l.knots.numel(), acts[i], sp) for i, l in enumerate(m.kan_layers)]
    import ctypes as C
    from rovit_hip.native import ptr_array
    arr = lambda xs: (C.c_int * len(xs))(*xs)
    gz = [torch.empty(B, layers[l + 1], device=dev) for l in range(n)]
    bargs = (ptr(x), ptr_array([l.spline_weights for l in m.kan_layers]), ptr_array([l.knots for l in m.kan_layers]),
             ptr_array([l.linear.weight for l in m.kan_layers]), ptr_array(outs), ptr_array([None, None, gout]), ptr_array(gz), ptr(dxs[0]),
             ptr_array(dws), ptr_array(dlw), ptr_array(dlb), B, arr(layers), arr([l.knots.numel() for l in m.kan_layers]), arr(acts), n, sp)

    def run():
        for r in fwd:
            lib.rovit_kan_layer_fwd(*r)
        lib.rovit_kan_stack_bwd(*bargs)
    ms = _event_avg_ms(dev, run, iters, per_launch=False)
    w_bytes = sum(a * b * nb + a * b + b for a, b in zip(layers[:-1], layers[1:])) * 4
    act_bytes = B * (layers[0] + 2 * sum(layers[1:-1]) + layers[-1]) * 4
    alg = float(3 * w_bytes + 3 * act_bytes)
    useful = 3 * 2.0 * B * sum(a * b * 5 for a, b in zip(layers[:-1], layers[1:]))
    gbs = alg / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': 'KAN stack forward (3 launches) + backward (2 launches: rovit_kan_stack_bwd), kernels only',
            'num_knots': G, 'batch': B, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 5),
            'avg_us': round(ms * 1e3, 2), 'algorithmic_bytes': alg, 'traffic': None,
            'useful_fp32_tflops': round(useful / (ms * 1e-3) / 1e12, 3), 'useful_frac_of_fp32_peak': round(useful / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5)}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline():
    """The CPU oracle (a port of the reference's algorithm; the reference itself needs timm and never travels to this
    box) on BOUNDED samples of the same workload, fp32, on this box's host cores -- BASELINE.md section 3's plan:
    backbone fwd and fwd+bwd at batch 256, the KAN head vectorised AND loop-faithful (what the reference's Python
    really does, models/kan.py:85-89), the full training step (one step at batch 256), and the reference's own fps() protocol as
    written (evaluation/metrics.py:63-93: batch 1, 10 warm-up + 100 timed forwards).  About 60-70 s in all."""
    from oracle import ref_cpu
    cores = min(len(os.sched_getaffinity(0)), 32)      # the box's CPU share, not the host's hardware threads
    torch.set_num_threads(cores)
    sd = ref_cpu.init_rovit_state(seed=0)
    out = {'cores': cores, 'cpu_model': _cpu_model(), 'kind': 'port', 'unit': 'images/sec'}

    def timed(fn, budget, max_n):
        fn()
        t0, n = time.time(), 0
        while n < max_n and (n == 0 or time.time() - t0 < budget):
            fn()
            n += 1
        return n, time.time() - t0

    # full training step (the quantity `value` is): fwd + loss + bwd.  Batch 16 warms the thread pool up and gives a several-sample
    # rate; then ONE step at the benchmark's own batch 256 (bounded: ~10-20 s), which is the figure `value` quotes.
    params = {k: (v.clone().requires_grad_(True) if 'knots' not in k else v) for k, v in sd.items()}
    B = 16
    x = torch.randn(B, 3, 224, 224)
    y = torch.randint(0, 4, (B,))

    def step():
        ref_cpu.joint_loss(ref_cpu.rovit_forward(x, params, 4), y, y, 4)['total_loss'].backward()
    n, dt = timed(step, 6.0, 20)
    out['train_step_b16'] = round(n * B / dt, 2)
    x256 = torch.randn(256, 3, 224, 224)
    y256 = torch.randint(0, 4, (256,))
    t0 = time.time()
    ref_cpu.joint_loss(ref_cpu.rovit_forward(x256, params, 4), y256, y256, 4)['total_loss'].backward()
    dt256 = time.time() - t0
    out['value'] = round(256 / dt256, 2)
    out['sample'] = (f'ONE training step (fwd + JointLoss + bwd) at batch 256 in {dt256:.1f} s, after {n} steps of batch {B} '
                     f'({out["train_step_b16"]} img/s); vectorised KAN restatement (oracle/ref_cpu.py), fp32, {cores} threads')
    del x256, y256
    # backbone alone at batch 256 (BASELINE.json configs[1])
    xb = torch.randn(256, 3, 224, 224)
    with torch.no_grad():
        n, dt = timed(lambda: ref_cpu.vit_forward(xb, sd, prefix='backbone.model.'), 4.0, 3)
    out['backbone_fwd_b256'] = round(n * 256 / dt, 1)
    bp = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('backbone.')}
    n, dt = timed(lambda: ref_cpu.vit_forward(xb, bp, prefix='backbone.model.').square().mean().backward(), 6.0, 2)
    out['backbone_fwd_bwd_b256'] = round(n * 256 / dt, 1)
    # KAN head, features (256,192): vectorised einsum restatement and the reference's in x out Python loop
    f = torch.randn(256, 192)
    with torch.no_grad():
        n, dt = timed(lambda: ref_cpu.kan_module_forward(f, sd, 'kan_module.'), 1.0, 50)
        out['kan_fwd_b256_vectorised_ms'] = round(dt / n * 1e3, 2)
        n, dt = timed(lambda: ref_cpu.kan_module_forward(f, sd, 'kan_module.', loop=True), 4.0, 2)
        out['kan_fwd_b256_loop_faithful_ms'] = round(dt / n * 1e3, 1)
        # the reference's fps() protocol AS WRITTEN (evaluation/metrics.py:63-93): batch 1, 10 warm-up forwards, 100 timed,
        # wall clock; stage 4 (KAN active).  Vectorised restatement and the loop-faithful KAN (what the reference's Python does).
        x1 = torch.randn(1, 3, 224, 224)

        def fps(fn, warm=10, iters=100):
            for _ in range(warm):
                fn()
            t0 = time.time()
            for _ in range(iters):
                fn()
            return iters / (time.time() - t0)

        def one_loop():
            feats = ref_cpu.vit_forward(x1, sd, prefix='backbone.model.')
            o = ref_cpu.heads_forward(feats, sd, 4)
            o['kan_severity'] = ref_cpu.kan_module_forward(feats, sd, 'kan_module.', loop=True)
        out['fps_protocol_batch1_kan_vectorised'] = round(fps(lambda: ref_cpu.rovit_forward(x1, sd, 4)), 2)
        out['fps_protocol_batch1_kan_loop'] = round(fps(one_loop), 2)
        out['fps_protocol'] = '10 warm-up + 100 timed forwards of a (1,3,224,224) tensor, wall clock (evaluation/metrics.py:63-93)'
    out['note'] = ('reference publishes 2.6 img/s (KAN active) / 36.7 img/s (backbone + cls head) for its fps() protocol on an '
                   'unnamed CPU (README.md:315,340)')
    return out


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def launcher_command(n, argv, port):
    """The command `bench.py --gpus N` starts when it was NOT itself started by a launcher: one rank per GPU on this node,
    rendezvous on 127.0.0.1 (the container hostname may not resolve), same script, same arguments."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def _fail(msg):
    print('bench.py: ' + msg, file=sys.stderr, flush=True)
    sys.exit(2)


def launch_or_check_world(args, argv):
    """Make `--gpus N` mean N ranks, loudly.  Called BEFORE anything touches the GPU (torch.cuda.device_count() does not
    initialise it on this image), so the parent of a self-launched job never holds a device:
      * launched by torch.distributed.run (WORLD_SIZE set): --gpus must equal WORLD_SIZE, and this rank's device must exist;
      * not launched and --gpus N > 1: start `python -m torch.distributed.run ... bench.py <same args>` as a CHILD process
        (never an exec), pass its output through (rank 0 prints the one JSON line) and exit with its return code;
      * fewer than N visible devices: exit 2 with a message -- never a line that says n_gpus 1 for --gpus 8.
    Returns only in the process that should run the benchmark itself."""
    gloo_selftest = args.selftest_gloo
    ws = os.environ.get('WORLD_SIZE')
    ndev = torch.cuda.device_count() if not gloo_selftest else args.gpus
    if ws is not None:
        if int(ws) != args.gpus:
            _fail(f'--gpus {args.gpus} disagrees with WORLD_SIZE={ws} set by the launcher; pass --gpus {ws} '
                  f'(or --nproc-per-node {args.gpus})')
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if local >= ndev:
            _fail(f'LOCAL_RANK {local} has no device: {ndev} GPU(s) visible, {ws} ranks requested')
        return
    if args.gpus < 1:
        _fail(f'--gpus must be >= 1 (got {args.gpus})')
    if args.gpus == 1:
        if ndev < 1:
            _fail('no GPU visible (torch.cuda.device_count() == 0): the benchmark has no CPU fallback')
        return
    if ndev < args.gpus:
        _fail(f'--gpus {args.gpus} requested but only {ndev} GPU(s) visible; refusing to report a {args.gpus}-GPU figure '
              f'from fewer devices')
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = launcher_command(args.gpus, argv, _free_port())
    print('bench.py: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f'bench.py: the {args.gpus}-rank job failed with exit code {rc}', file=sys.stderr, flush=True)
    sys.exit(rc)


def _selftest_gloo(args):
    """Launcher rehearsal without a GPU (tests/test_bench_launcher.py): every rank joins a gloo group, one all-reduce, rank 0
    prints a JSON line with the world it saw.  Exercises the self-launch, the environment hand-over and the exit code."""
    world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
    dist.init_process_group('gloo')
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({'selftest': 'gloo', 'n_gpus': world, 'sum_of_ranks_plus_one': float(t.item()), 'gpus_arg': args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if args.selftest_fail_rank == rank:
        sys.exit(7)


def _comm_report(sync, dev, step, iters=10):
    """What the exchange costs, measured after the timed region on rank 0's clock (all ranks run it: collectives):
    per-bucket sizes, each bucket's all-reduce ALONE (device events, nothing else running), and the time the main stream
    spends in GradSync.finish() inside real steps = the part of the exchange the backward did not hide."""
    rep = {'buckets': []}
    eng = sync.engine
    slices = [sync.slice_for(f, l) for f, l in sync.ranges]
    st = torch.cuda.current_stream(dev)
    for (f, l), (off, n) in zip(sync.ranges, slices):
        piece = eng.grad_flat[off:off + n]
        for _ in range(3):
            dist.all_reduce(piece, op=dist.ReduceOp.SUM)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            dist.all_reduce(piece, op=dist.ReduceOp.SUM)
        e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        rep['buckets'].append({'blocks': [f, l], 'bytes': 4 * n, 'allreduce_alone_us': round(us, 1),
                               'bus_GBps': round(2 * (sync.world - 1) / sync.world * 4 * n / (us * 1e-6) / 1e9, 1)})
    evs = []
    orig_finish = sync.finish

    def timed_finish():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        orig_finish()
        b.record(st)
        evs.append((a, b))
    sync.finish = timed_finish
    try:
        for _ in range(iters):
            step()
    finally:
        sync.finish = orig_finish
    torch.cuda.synchronize(dev)
    rep['exposed_allreduce_us_per_step'] = round(sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3, 1)
    rep['note'] = ('exposed = device time between the events around GradSync.finish() in real steps (main stream waiting for the '
                   'reduction stream); alone = each bucket all-reduced by itself, nothing else on the device')
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--buckets', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true',
                    help='time the step only (the command profiles/r03_bench_kernel_stats.csv is taken from: its kernel times sum to the step)')
    ap.add_argument('--roofline-only', action='store_true',
                    help='run only the roofline kernel measurements (the command profiles/r02_roofline_kernel_stats.csv and the PMC passes are taken from)')
    ap.add_argument('--selftest-gloo', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--selftest-fail-rank', type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.roofline_only:
        dev = torch.device('cuda:0')
        _warm_clocks(dev)
        print(json.dumps({'roofline': wgrad_roofline(dev), 'roofline_attn': attn_roofline(dev), 'roofline_mlp': mlp_roofline(dev),
                          'roofline_kan': kan_roofline(dev), 'roofline_patch': patch_roofline(dev)}), flush=True)
        return

    launch_or_check_world(args, sys.argv[1:])            # returns only in a process that runs the benchmark itself
    if args.selftest_gloo:
        return _selftest_gloo(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus
    force_dist = os.environ.get('ROVIT_FORCE_DIST') == '1'      # exercise the RCCL code path on a single GPU
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:                  # single-process RCCL check: any free port
            os.environ['MASTER_PORT'] = str(_free_port())
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local)
        import datetime
        # a collective that never completes must end the run with an error, not hold an 8-GPU lease until the driver's limit
        dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=datetime.timedelta(minutes=4))
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
    sync = GradSync(model, buckets=args.buckets, force=force_dist, optimizer=opt)
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
    rccl_world = dist.get_world_size() if dist.is_initialized() else 0
    if world > 1 and rccl_world != args.gpus:
        _fail(f'the RCCL group has {rccl_world} ranks, --gpus says {args.gpus}')
    comm = _comm_report(sync, dev, step) if (world > 1 or force_dist) else None

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
                       'rccl_world': rccl_world,
                       'backbone_mfma_frac_of_step': round(ips / world * TRAIN_FLOP_PER_IMG / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
            'final_loss': round(final_loss, 5),
        }
        if comm is not None:
            res['config']['allreduce'] = comm
            res['config']['allreduce_op'] = sync.reducer.op_name
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()       # the other ranks are done: rank 0 measures its kernels alone
    if rank == 0:
        # N > 1 lines carry the step and the exchange only (the kernels are the same as at N = 1; eight-rank runs stay short)
        if not args.no_roofline and world == 1:
            res['roofline'] = wgrad_roofline(dev)
            res['roofline_attn'] = attn_roofline(dev)
            res['roofline_mlp'] = mlp_roofline(dev)
            res['roofline_kan'] = kan_roofline(dev)
            res['fp32_mode'] = fp32_mode(dev)
            res['roofline_patch'] = patch_roofline(dev)
            res['backbone_only'] = backbone_only(dev)
            attn_us = sum(v['avg_us'] for v in res['roofline_attn'].values())
            attn_floor_us = sum(v['algorithmic_bytes'] for v in res['roofline_attn'].values()) / (HBM_PEAK_GBS * 1e9) * 1e6
            # images/sec as a fraction of the attention-GEMM roofline (north_star): the six attention kernels of one block
            # handle 256 images; their HBM-roofline time on algorithmic bytes against their measured time
            res['config']['attention_half_us_per_block'] = round(attn_us, 1)
            res['config']['attention_half_frac_of_hbm_roofline'] = round(attn_floor_us / attn_us, 4)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
