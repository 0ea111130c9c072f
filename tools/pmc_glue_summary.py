#!/usr/bin/env python3
"""FETCH_SIZE (x2, gfx950 correction of MI355X_MICROARCH.md) / WRITE_SIZE and duration per launch of the depthwise and InstanceNorm
kernels from the two passes of tools/pmc_glue.sh, next to the algorithmic bytes.  Usage: pmc_glue_summary.py <fetch dir> <write dir>"""
import csv, glob, os, sys
csv.field_size_limit(1 << 30)


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value']), int(r.get('Grid_Size', 0) or 0)))
    return sorted(rows)


def dur(d):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return out


fetch, write, durs = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE'), dur(sys.argv[1])
names = ['depthwise_lds_kernel', 'depthwise_fwd_kernel', 'depthwise_wgrad_kernel', 'cnorm_stats_kernel<0>', 'cnorm_apply_kernel<0>', 'cnorm_stats_kernel<1>', 'cnorm_apply_kernel<1>']
MB = {48: 2 * 48 ** 3 * 256 * 4 / 1e6, 24: 2 * 24 ** 3 * 512 * 4 / 1e6}
alg = {'depthwise_lds_kernel': 2, 'depthwise_fwd_kernel': 2, 'depthwise_wgrad_kernel': 2, 'cnorm_stats_kernel<0>': 1, 'cnorm_apply_kernel<0>': 2, 'cnorm_stats_kernel<1>': 2,
       'cnorm_apply_kernel<1>': 3}
print('# rocprofv3 PMC, attention-stage kernels (tools/pmc_glue.sh): HBM-side bytes per launch vs algorithmic bytes\n')
print('FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-byte requests at 64 B); KB -> MB.  Shapes (2,48,48,48,256) = 226 MB and '
      '(2,24,24,24,512) = 57 MB per tensor, fp32.  depthwise_lds_kernel (48^3) / depthwise_fwd_kernel (24^3) are also the data-gradient kernels (every second launch).\n')
print('| kernel | tensor MB | algorithmic MB | FETCH x2 MB | WRITE MB | us | achieved GB/s (algorithmic) |')
print('|---|---|---|---|---|---|---|')
for n in names:
    f = [r for r in fetch if n in r[1]]
    w = [r for r in write if n in r[1]]
    d = [v for k, vs in durs.items() if n in k for v in vs]
    if not f:
        continue
    half = len(f) // 2
    only = {'depthwise_lds_kernel': 48, 'depthwise_fwd_kernel': 24}.get(n) if any('depthwise_lds_kernel' in r[1] for r in fetch) else None
    for tag, sl in (((only, slice(0, None)),) if only else ((48, slice(0, half)), (24, slice(half, None)))):
        ff, ww, dd = f[sl], w[sl], d[sl]
        if not ff:
            continue
        fm = sum(r[2] for r in ff) / len(ff) * 2 / 1e3
        wm = sum(r[2] for r in ww) / max(len(ww), 1) / 1e3
        us = sum(dd) / max(len(dd), 1)
        a = alg[n] * MB[tag]
        print(f'| {n} | {MB[tag]:.0f} | {a:.0f} | {fm:.0f} | {wm:.0f} | {us:.1f} | {a / us * 1e3:.0f} |')
