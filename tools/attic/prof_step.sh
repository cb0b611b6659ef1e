#!/bin/bash
# rocprofv3 kernel-trace summary of the benchmark step: two-stream (default) and single-stream (serial kernel durations).
# usage (on the GPU box): tools/prof_step.sh <tag>   -> gpurun_out/prof_<tag>{,_ss}/..._kernel_stats.csv
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
ROVIT_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_ss -o ${TAG}_ss -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_${TAG}_ss.log 2>&1
python3 - <<PY
import csv, glob
for tag in ("$TAG", "${TAG}_ss"):
    f = glob.glob("$OUT/prof_%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(tag, 'total kernel ms per step', tot / 25 / 1e6)
    for r in rows[:14]:
        print('  %-70s calls/step %6.1f avg_us %8.2f  %5.1f%%' % (r['Name'][:70], int(r['Calls']) / 25, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
