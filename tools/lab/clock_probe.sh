#!/bin/bash
# clock and matrix-pipe occupancy per lab variant: SQ_BUSY_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES (PMC pass) against the durations of a kernel-trace pass
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out/lab
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- tools/lab/bin/f32_gemm_lab > $OUT/kt.log 2>&1 || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -o pmc -- tools/lab/bin/f32_gemm_lab > $OUT/pmc.log 2>&1 || exit 1
python3 - <<PY > $OUT/clock.txt
import csv, glob, collections
kt = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[(r['Kernel_Name'], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
pm = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
rows = list(csv.DictReader(open(pm)))
gx = 'Grid_Size_X' if 'Grid_Size_X' in rows[0] else None
for r in rows:
    key = (r['Kernel_Name'], r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', '')) if gx else (r['Kernel_Name'], r['Grid_Size'])
    cnt[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(dur):
    d = sorted(dur[k]); d = d[len(d) // 2]
    c = cnt.get(k) or cnt.get((k[0], str(int(k[1]) * int(k[2]))))
    if not c: print(k[0][-40:], k[1], k[2], 'median %.1f us (no counters)' % d); continue
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    busy = avg.get('SQ_BUSY_CYCLES', 0) / 32
    print('%-46s grid %5s x %4s  %7.1f us  clock %.2f GHz  matrix pipe busy %.2f of the kernel  waiting %.2f of wave-cycles' % (
        k[0][-46:], k[1], k[2], d, busy / d / 1e3, avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(busy, 1), avg.get('SQ_WAIT_INST_ANY', 0) / max(avg.get('SQ_WAVE_CYCLES', 1), 1)))
PY
rm -rf $OUT/kt $OUT/pmc
