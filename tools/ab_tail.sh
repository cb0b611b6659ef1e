#!/bin/bash
# A/B of the forward block tail: 8 waves x 2 tiles against 16 waves x 1 tile (developer library, knob 18)
L=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib/librovit_hip_dev.so
for k in 8 16; do
  echo "== tail waves $k"
  ROVIT_HIP_LIB=$L ROVIT_DEV_KNOBS=18=$k python -c "
import json, torch, bench
d = bench.mlp_roofline(torch.device('cuda:0'))
for k in ('block_tail_train', 'block_tail_inference'):
    print(k, d[k]['avg_us'], d[k]['frac'])
" 2>&1 | grep block_tail
done
echo "== parity of the 16-wave kernel (block-tail tests under the developer library)"
ROVIT_HIP_LIB=$L ROVIT_DEV_KNOBS=18=16 timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -q -k "block_tail_equals or tail" 2>&1 | tail -4
