#!/bin/bash
# rocprofv3 kernel-trace summary of `bench.py --roofline-only` (the measurement `roofline.avg_us` must agree with)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roof_r02 -o roof -- python3 bench.py --roofline-only > $OUT/prof_roof_r02.log 2>&1
grep -o '{"roofline.*' $OUT/prof_roof_r02.log | tail -1 > $OUT/roofline_only_under_rocprof.json || true
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/prof_roof_r02/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('wgrad_kernel', 'gemm_ws_dma', 'kan_')):
        print('%-90s calls %5s avg_us %9.2f' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
