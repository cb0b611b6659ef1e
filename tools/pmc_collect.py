"""PMC evidence for the roofline kernels (run on the GPU box): rocprofv3 --pmc passes over `python3 bench.py --roofline-only`,
as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, nothing combined with trace domains, read
bytes = 2 x FETCH_SIZE on gfx950 -- plus two SQ passes (matrix / vector / LDS activity) for the kernels whose bound is argued in
DESIGN.md.  Writes gpurun_out/pmc_r04/r04_pmc_traffic.json and r04_pmc_sq.json (copy them to profiles/).

    python3 tools/pmc_collect.py [--skip-sq]

This script itself never touches the GPU; the profiled program is started directly behind `--` (no shell, no env wrapper).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'pmc_r04')

# (substring of rocprof's Kernel_Name, occurrence group ordered by first dispatch when the same kernel runs two shapes, key)
KERNELS = [
    ('wgrad_kernel<192, 192, false, false, 4>', 0, 'wgrad_kernel<192,192>'),
    ('gemm_ws_dma_kernel<0>', 0, 'qkv_fwd'),
    ('gemm_ws_dma_kernel<0>', 1, 'proj_dgrad'),
    ('attn_fwd_kernel', 0, 'attention_fwd'),
    ('attn_bwd_kernel', 0, 'attention_bwd'),
    ('gemm_ws_kernel<6, 1, 64, 5>', 0, 'proj_fwd_resid_ln'),
    ('gemm_kdma_kernel<18, 6>', 0, 'qkv_dgrad_ln_bwd'),
    ('mlp_fused_kernel<0, 2, 8, false, true, true,', 0, 'block_tail_train'),
    ('mlp_fused_kernel<0, 0, 8, false, true, true,', 0, 'block_tail_inference'),
    ('mlp_fused_kernel<0, 2, 8, false, true, false,', 0, 'mlp_fused_fwd_train'),
    ('mlp_fused_kernel<0, 0, 8, false, true, false,', 0, 'mlp_fused_fwd_inference'),
    ('mlp_fused_kernel<1, 1, 8, false, false, false,', 0, 'mlp_fused_bwd'),
    ('kan_fwd_kernel', 0, 'kan_fwd_kernel'),
    ('gemm_ws_kernel<12, 2, 32, 7>', 0, 'patch_embed_fwd'),
    ('wgrad_kernel<192, 96, true, true, 2>', 0, 'patch_embed_wgrad'),
    ('kan_stack_mfma_kernel<4, 4', 0, 'kan_stack_mfma_kernel<4>'),
    ('kan_stack_mfma_kernel<18, 1', 0, 'kan_stack_mfma_kernel<18>'),
    ('head_phase_fwd_kernel', 0, 'head_phase_fwd'),
    ('head_phase_bwd_dx_kernel', 0, 'head_phase_bwd_per_sample'),
    ('head_phase_dw_kernel', 0, 'head_phase_bwd_params'),
]
SQ_PASSES = [
    ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU',
     'SQ_INSTS_MFMA'],
    ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS',
     'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_VALU_MFMA_MOPS_BF16'],
]


def run_pass(tag, counters):
    d = os.path.join(OUT, tag)
    cmd = ['rocprofv3', '--pmc', *counters, '--output-format', 'csv', '-d', d, '-o', tag, '--', sys.executable, os.path.join(ROOT, 'bench.py'),
           '--roofline-only']
    env = dict(os.environ, TMPDIR='/tmp')
    with open(os.path.join(OUT, tag + '.log'), 'w') as log:
        rc = subprocess.call(cmd, cwd='/tmp', env=env, stdout=log, stderr=subprocess.STDOUT)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return rc, None
    # keep ONE compact copy next to the logs and drop rocprofv3's output tree (gpurun merges at most 64 MiB back)
    keep = os.path.join(OUT, tag + '_counter_collection.csv')
    cols = ('Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Counter_Name', 'Counter_Value')
    with open(files[0]) as fin, open(keep, 'w', newline='') as fout:
        wr = csv.writer(fout)
        wr.writerow(cols)
        for r in csv.DictReader(fin):
            if any(sub in r['Kernel_Name'] for sub, _, _ in KERNELS):
                wr.writerow([r.get('Dispatch_Id', ''), r['Kernel_Name'], r.get('Grid_Size', ''), r['Counter_Name'], r['Counter_Value']])
    shutil.rmtree(d, ignore_errors=True)
    return rc, keep


def per_kernel(path, counter):
    """{key: (mean, launches)} of one counter; a kernel that runs two shapes is split by grid size, groups ordered by first dispatch."""
    groups = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        groups.setdefault((r['Kernel_Name'], r.get('Grid_Size', '')), []).append(float(r['Counter_Value']))
    out = {}
    for sub, idx, key in KERNELS:
        g = [(k, v) for k, v in groups.items() if sub in k[0]]
        if idx < len(g):
            v = g[idx][1]
            out[key] = (sum(v) / len(v), len(v))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    res = {}
    rc_f, f_csv = run_pass('fetch', ['FETCH_SIZE'])
    rc_w, w_csv = run_pass('write', ['WRITE_SIZE'])
    if f_csv and w_csv:
        fe, wr = per_kernel(f_csv, 'FETCH_SIZE'), per_kernel(w_csv, 'WRITE_SIZE')
        for _, _, key in KERNELS:
            if key in fe and key in wr:
                f_kb, n1 = fe[key]
                w_kb, n2 = wr[key]
                res[key] = {'FETCH_SIZE_KB_mean': round(f_kb, 1), 'WRITE_SIZE_KB_mean': round(w_kb, 1), 'launches': [n1, n2],
                            'traffic_bytes': round((2 * f_kb + w_kb) * 1024), 'note': 'read bytes = 2 x FETCH_SIZE (gfx950), counters in KB'}
        # the per-layer KAN forward at the reference's own sizes: bench.py --roofline-only runs C5 (G = 32, batch 512) first and C3 (G = 5,
        # batch 256) last, three kan_fwd_kernel launches each with their own grid: a configuration's traffic = the sum over its layers
        def kan_groups(path, counter):
            groups = collections.OrderedDict()
            for r in csv.DictReader(open(path)):
                if r['Counter_Name'] == counter and 'kan_fwd_kernel' in r['Kernel_Name']:
                    groups.setdefault(r.get('Grid_Size', ''), []).append(float(r['Counter_Value']))
            return [sum(v) / len(v) for v in groups.values()]
        gf, gw = kan_groups(f_csv, 'FETCH_SIZE'), kan_groups(w_csv, 'WRITE_SIZE')
        if len(gf) == 6 and len(gw) == 6:
            for key, sl in (('kan_fwd_c5_g32_b512', slice(0, 3)), ('kan_fwd_c3_g5_b256', slice(3, 6))):
                res[key] = {'FETCH_SIZE_KB_sum_of_3_layers': round(sum(gf[sl]), 1), 'WRITE_SIZE_KB_sum_of_3_layers': round(sum(gw[sl]), 1),
                            'traffic_bytes': round((2 * sum(gf[sl]) + sum(gw[sl])) * 1024), 'note': 'three per-layer launches, read bytes = 2 x FETCH_SIZE'}
        else:
            res['kan_fwd_per_layer_note'] = 'expected 6 kan_fwd_kernel grid groups (C5 then C3), found %d / %d' % (len(gf), len(gw))
    json.dump(res, open(os.path.join(OUT, 'r04_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps({k: v['traffic_bytes'] for k, v in res.items() if isinstance(v, dict)}, indent=1), 'rc', rc_f, rc_w)
    if '--skip-sq' in sys.argv:
        return
    sq = {}
    for i, counters in enumerate(SQ_PASSES):
        rc, path = run_pass(f'sq{i}', counters)
        if not path:
            sq[f'pass{i}_error'] = f'rocprofv3 rc {rc}: see gpurun_out/pmc_r04/sq{i}.log (a counter name may not exist on this ROCm)'
            continue
        for c in counters:
            for key, (mean, n) in per_kernel(path, c).items():
                sq.setdefault(key, {})[c] = round(mean, 1)
    # derived shares (SQ counters are summed over the chip; WAVE_CYCLES etc. count quad-cycles: ratios only)
    for key, v in sq.items():
        if not isinstance(v, dict):
            continue
        wc = v.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS'):
                if c in v:
                    v[c + '_share_of_wave_cycles'] = round(v[c] / wc, 4)
        if v.get('SQ_LDS_IDX_ACTIVE'):
            v['lds_bank_conflict_share_of_lds_cycles'] = round(v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE'], 4)
    json.dump(sq, open(os.path.join(OUT, 'r04_pmc_sq.json'), 'w'), indent=1)
    print(json.dumps(sq, indent=1)[:3000])


if __name__ == '__main__':
    main()
