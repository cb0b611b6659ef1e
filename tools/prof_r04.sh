#!/bin/bash
# rocprofv3 kernel-trace summaries of round 4 (run on the GPU box):
#   1. the benchmark step alone (bench.py --no-roofline: the kernel times of this profile sum to the step), two streams (product library)
#   2. the same on ONE stream (developer library, knob 4: serial per-kernel durations)
#   3. bench.py --roofline-only (what `roofline*.avg_us` must agree with)
# usage: tools/prof_r04.sh [tag]   -> gpurun_out/prof_<tag>{,_ss,_roof}/ + gpurun_out/<tag>_bench_summary.txt
set -e
TAG=${1:-r04}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$ROOT/gpurun_out
DEVLIB=$ROOT/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib/librovit_hip_dev.so
STEPS=20; WARM=5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline > $OUT/prof_$TAG.log 2>&1
ROVIT_HIP_LIB=$DEVLIB ROVIT_DEV_KNOBS=4=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_ss -o ${TAG}_ss -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline > $OUT/prof_${TAG}_ss.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_roof -o ${TAG}_roof -- python3 bench.py --roofline-only > $OUT/prof_${TAG}_roof.log 2>&1
grep -o '{"roofline.*' $OUT/prof_${TAG}_roof.log | tail -1 > $OUT/${TAG}_roofline_only_under_rocprof.json || true
python3 - <<PY > $OUT/${TAG}_bench_summary.txt
import csv, glob, json
n = $STEPS + $WARM
for tag in ("$TAG", "${TAG}_ss"):
    f = glob.glob("$OUT/prof_%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    line = [l for l in open("$OUT/prof_%s.log" % tag) if l.startswith('{"metric"')]
    ms = json.loads(line[-1])['ms_per_step'] if line else None
    print('%s: sum of kernel durations per step %.3f ms (%d steps incl. warm-up); bench.py ms_per_step under the profiler: %s' % (tag, tot / n / 1e6, n, ms))
    for r in rows[:24]:
        print('  %-86s calls/step %6.1f avg_us %8.2f  %5.1f%%' % (r['Name'][:86], int(r['Calls']) / n, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
f = glob.glob("$OUT/prof_${TAG}_roof/**/*kernel_stats.csv", recursive=True)[0]
print('roofline-only (bench.py --roofline-only): rocprofv3 average per kernel')
for r in csv.DictReader(open(f)):
    print('  %-86s calls %5s avg_us %9.2f' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3))
PY
cat $OUT/${TAG}_bench_summary.txt
for t in $TAG ${TAG}_ss ${TAG}_roof; do cp $(find $OUT/prof_$t -name '*kernel_stats.csv' | head -1) $OUT/${t}_kernel_stats.csv; rm -rf $OUT/prof_$t; done
