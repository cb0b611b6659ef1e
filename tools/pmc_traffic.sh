#!/bin/bash
# HBM traffic of the roofline kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (no trace domains combined with --pmc), read bytes = 2 x FETCH_SIZE on gfx950.
# usage (GPU box): tools/pmc_traffic.sh  -> profiles/r02_pmc_traffic.json (+ the raw per-kernel means in gpurun_out/pmc_r02/)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=$PWD/gpurun_out/pmc_r02
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 bench.py --roofline-only > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 bench.py --roofline-only > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob, json, re, collections
def means(d, counter):
    f = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe, wr = means('fetch', 'FETCH_SIZE'), means('write', 'WRITE_SIZE')
out = {}
for name, key in (('wgrad_kernel<96, 192, false, false>', 'wgrad_kernel<96,192,false>'), ('gemm_ws_dma_kernel<1>', 'gemm_ws_dma_kernel<1>'),
                  ('kan_stack_fwd_kernel<32>', 'kan_stack_fwd_kernel<32>'), ('kan_fwd_kernel', 'kan_fwd_kernel'),
                  ('kan_stack_mfma_kernel<4, 4>', 'kan_stack_mfma_kernel<4>'), ('kan_stack_mfma_kernel<18, 1>', 'kan_stack_mfma_kernel<18>')):
    fk = [k for k in fe if name in k]; wk = [k for k in wr if name in k]
    if not fk or not wk:
        continue
    f_kb, n1 = fe[fk[0]]; w_kb, n2 = wr[wk[0]]
    out[key] = {'FETCH_SIZE_KB_mean': round(f_kb, 1), 'WRITE_SIZE_KB_mean': round(w_kb, 1), 'launches': [n1, n2],
                'traffic_bytes': round((2 * f_kb + w_kb) * 1024), 'note': 'read bytes = 2 x FETCH_SIZE (gfx950), counters in KB'}
json.dump(out, open("$OUT/r02_pmc_traffic.json", "w"), indent=1)   # copy to profiles/ after the run (only gpurun_out/ travels back)
print(json.dumps(out, indent=1))
PY
