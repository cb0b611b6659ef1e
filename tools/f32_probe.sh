#!/bin/bash
# fp32 reference-precision forward, batch 256, ONE stream (developer library, knob 4): per-kernel durations (rocprofv3 --kernel-trace --stats)
# and, in passes of their own, the matrix-pipe / LDS counters of its kernels.   usage: tools/f32_probe.sh  -> gpurun_out/f32_probe/
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out/f32_probe
mkdir -p $OUT
export ROVIT_HIP_LIB=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib/librovit_hip_dev.so
export ROVIT_DEV_KNOBS=${KNOBS:-4=1}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 tools/time_fp32.py > $OUT/kt.log 2>&1 || exit 1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 - <<PY > $OUT/b256_kernels.txt
import csv, glob, collections
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the batch-256 launches are the ones with the larger grids: group by (name, grid)
acc = collections.defaultdict(list)
for r in rows:
    acc[(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print('%-62s grid %8s x %5s  n %4d  median %8.2f us  min %8.2f' % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
PY
rm -rf $OUT/kt
if [ -n "$PMC" ]; then
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc -o pmc -- python3 tools/time_fp32.py > $OUT/pmc.log 2>&1 || exit 1
  python3 - <<PY > $OUT/b256_pmc.txt
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[(r['Kernel_Name'][:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(acc.items()):
    print(k[0], 'grid', k[1], ' '.join('%s=%.3g' % (n, sum(v) / len(v)) for n, v in sorted(c.items())))
PY
  rm -rf $OUT/pmc
  # HBM traffic of the same kernels: FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: KB units, read bytes = 2 x FETCH_SIZE on gfx950)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 tools/time_fp32.py 256 > $OUT/pmc_$c.log 2>&1 || exit 1
  done
  python3 - <<PY > $OUT/b256_traffic.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c and 'f32' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:64], r['Grid_Size'])][c].append(float(r['Counter_Value']))
M = 256 * 197
alg = {'921600': ('qkv', M * 192 * 4 + M * 576 * 4), '1228800': ('fc1', M * 192 * 4 + M * 768 * 4), '393216': ('attention', M * 576 * 4 + M * 192 * 4)}
for k, c in sorted(acc.items()):
    fe = sum(c['FETCH_SIZE']) / max(len(c['FETCH_SIZE']), 1); wr = sum(c['WRITE_SIZE']) / max(len(c['WRITE_SIZE']), 1)
    tr = (2 * fe + wr) * 1024
    name, a = alg.get(k[1], ('', 0))
    print('%-66s grid %8s  read %7.1f MB  written %7.1f MB  traffic %7.1f MB%s' % (k[0], k[1], 2 * fe * 1024 / 1e6, wr * 1024 / 1e6, tr / 1e6,
          ('  (%s: algorithmic %.1f MB, %.2fx)' % (name, a / 1e6, tr / a)) if a else ''))
PY
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
fi
