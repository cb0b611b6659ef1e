#!/bin/bash
# Kernel launch sequence of one training step (developer library, knob 4 = single stream, so that the trace order is the program order): rocprofv3 --kernel-trace,
# then the names of the launches of the LAST step in start order.  usage: tools/kernel_sequence.sh  -> gpurun_out/kernel_sequence.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out
ROVIT_HIP_LIB=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib/librovit_hip_dev.so ROVIT_DEV_KNOBS=4=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kseq -o kseq -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/kseq.log 2>&1
python3 - <<PY > $OUT/kernel_sequence.txt
import csv, glob
f = glob.glob("$OUT/kseq/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last step = behind the last AdamW launch but one
idx = [i for i, n in enumerate(names) if 'adamw_multi_kernel' in n]      # one AdamW launch per step (round 4)
lo = idx[-2] + 1 if len(idx) >= 2 else 0
prev_end = None
for r in rows[lo:idx[-1] + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%8.2f us  gap %6.2f  %s' % ((e - s) / 1e3, gap, r['Kernel_Name'][:110]))
    prev_end = e
PY
rm -rf $OUT/kseq
