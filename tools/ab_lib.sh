#!/bin/bash
# A/B of two builds of the library on one box: usage ab_lib.sh <libA> <libB>  (bench.py --steps 60, three alternations)
P=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib
for i in 1 2 3; do
  for l in "$@"; do
    printf "%s  " "$l"
    ROVIT_HIP_LIB=$P/$l python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
