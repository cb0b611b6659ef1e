"""What the gelu' round trip costs the FORWARD: the block-tail launch (the step's forward kernel) with act + gelu' kept (training,
MODE 2), with act only (MODE 1: gelu' would be recomputed by the backward) and with nothing kept (inference), interleaved
repetitions on one box, device events per launch.  VERDICT r3 item 7 (the backward half of the trade is argued in DESIGN.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from rovit_hip import native  # noqa: E402

dev = torch.device('cuda:0')
lib = native.load()
p, sp = native.ptr, native.stream_ptr()
M = 256 * 197
bf = torch.bfloat16
o = torch.randn(M, 192, device=dev).to(bf)
w1 = (torch.randn(768, 192, device=dev) * 0.08).to(bf)
w2 = (torch.randn(192, 768, device=dev) * 0.05).to(bf)
wp = (torch.randn(192, 192, device=dev) * 0.05).to(bf)
wq = (torch.randn(576, 192, device=dev) * 0.05).to(bf)
bp, b1, b2, bq = (torch.randn(n, device=dev) * 0.2 for n in (192, 768, 192, 576))
X = torch.randn(M, 192, device=dev)
ws = torch.empty(lib.rovit_mlp_stream_bytes(), dtype=torch.uint8, device=dev)
native.call('rovit_mlp_prepare_stream_tail', p(w1), p(w2), p(wp), p(wq), p(ws), sp)
# rotate through several output sets so that the stores do not land in the Infinity Cache of the previous launch
sets = [dict(xh2=torch.empty(M, 192, device=dev, dtype=bf), r2=torch.empty(M, device=dev), act=torch.empty(M, 768, device=dev, dtype=bf),
             dact=torch.empty(M, 768, device=dev, dtype=bf), xh=torch.empty(M, 192, device=dev, dtype=bf), r=torch.empty(M, device=dev),
             qkv=torch.empty(M, 576, device=dev, dtype=bf)) for _ in range(6)]
it = [0]


def run(mode):
    s = sets[it[0] % len(sets)]
    it[0] += 1
    act = p(s['act']) if mode >= 1 else None
    dact = p(s['dact']) if mode == 2 else None
    lib.rovit_block_tail_fwd(p(o), p(ws), p(bp), p(b1), p(b2), p(X), p(s['xh2']) if mode else None, p(s['r2']) if mode else None, act, dact,
                             p(s['xh']), p(s['r']), p(bq), p(s['qkv']), 1e-6, M, M, sp)


bench._warm_clocks(dev)
res = {'note': 'block tail forward, M = 50432, six rotating output sets; us per launch (device events around every launch)'}
for rep in range(3):
    for mode, name in ((2, 'train_act_and_gelu_grad_kept'), (1, 'act_only_kept'), (0, 'inference_nothing_kept')):
        res.setdefault(name, []).append(round(bench._event_avg_ms(dev, lambda: run(mode), 30) * 1e3, 2))
res['bytes_written_per_launch_MB'] = {'train_act_and_gelu_grad_kept': round((2 * M * 768 * 2 + M * 192 * 2 * 2 + M * 576 * 2 + M * 192 * 4) / 1e6, 1),
                                      'act_only_kept': round((M * 768 * 2 + M * 192 * 2 * 2 + M * 576 * 2 + M * 192 * 4) / 1e6, 1)}
print(json.dumps(res))
