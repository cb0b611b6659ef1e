"""Merged weight-gradient launch: standalone time per (tile, splits).  ROVIT_WGRAD_MERGE_TN selects the tile (64 / 96 / 192).
python tools/exp_wgrad3.py 12 14 16 18 21 24"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch
from rovit_hip import native
lib = native.load()
dev = torch.device('cuda:0')
M = 256 * 197
shapes = [(576, 192), (192, 768), (768, 192), (192, 192)]
dY = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n, _ in shapes]
A = [torch.randn(M, k, device=dev).to(torch.bfloat16) for _, k in shapes]
arr = lambda xs: (C.c_int * len(xs))(*xs)
a_dy, a_a = native.ptr_array(dY), native.ptr_array(A)
ldy, lda, Ns, Ks = arr([n for n, _ in shapes]), arr([k for _, k in shapes]), arr([n for n, _ in shapes]), arr([k for _, k in shapes])
st = torch.cuda.current_stream(dev)
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(50): a @ a
out = {'tile_tn': os.environ.get('ROVIT_WGRAD_MERGE_TN', 'default')}
for S in [int(x) for x in sys.argv[1:]] or [16]:
    ws = [torch.empty(lib.rovit_wgrad_workspace_bytes(n, k, S), dtype=torch.uint8, device=dev) for n, k in shapes]
    a_ws = native.ptr_array(ws)
    run = lambda: native.call('rovit_wgrad_multi', a_dy, ldy, a_a, lda, Ns, Ks, a_ws, 4, M, S, native.stream_ptr())
    for _ in range(5): run()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for e0, e1 in evs:
        e0.record(st); run(); e1.record(st)
    evs[-1][1].synchronize()
    out[S] = round(sum(e0.elapsed_time(e1) for e0, e1 in evs) / 30 * 1e3, 2)
print(json.dumps(out))
