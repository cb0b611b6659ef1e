#!/bin/bash
# kernel sequence (duration, gap to the previous kernel's end) of the LAST batch-1 inference forward of a loop -> gpurun_out/kernel_sequence_b1.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $OUT/kseq1 -o kseq1 -- python3 tools/fwd_b1_trace.py ${1:-1} 30 > $OUT/kseq1.log 2>&1
python3 - <<PY > $OUT/kernel_sequence_b1.txt
import csv, glob
f = glob.glob("$OUT/kseq1/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'cls_row_kernel' in n or 'patch' in n.lower()]
starts = [i for i, n in enumerate(names) if 'cls_row_kernel' in n]
lo = starts[-1] if starts else 0
prev_end = None; tot = 0; gaps = 0
for r in rows[lo:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    tot += (e - s) / 1e3; gaps += max(gap, 0)
    print('%8.2f us  gap %6.2f  %s' % ((e - s) / 1e3, gap, r['Kernel_Name'][:100]))
    prev_end = e
print('kernels', len(rows) - lo, 'sum of durations us', round(tot, 1), 'sum of gaps us', round(gaps, 1))
PY
rm -rf $OUT/kseq1
