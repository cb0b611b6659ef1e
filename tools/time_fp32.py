"""fp32 reference-precision forward at batch 256: ms per forward and the per-kernel split (device events around one block's launches
are not available from Python, so the split comes from rocprofv3: `rocprofv3 --kernel-trace --stats -- python3 tools/time_fp32.py`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda:0')
bench._warm_clocks(dev)
batches = [int(a) for a in sys.argv[1:]] or [64, 256]
print(json.dumps({b: bench.fp32_mode(dev, b) for b in batches}))
