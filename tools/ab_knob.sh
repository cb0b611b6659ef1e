#!/bin/bash
# A/B of one developer-library knob inside whole training steps on one box: usage ab_knob.sh <knob id> <value> [<value> ...]
# (bench.py --steps 60, the value list run twice in alternation)
L=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib/librovit_hip_dev.so
K=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    printf "knob %s = %s   " "$K" "$v"
    ROVIT_HIP_LIB=$L ROVIT_DEV_KNOBS=$K=$v python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
