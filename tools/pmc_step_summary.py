#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per kernel family of the bench step (tools/pmc_step.sh) -> markdown on stdout and
profiles/conv_traffic.json (read by bench.py for roofline.traffic).  FETCH_SIZE is doubled (gfx950 tallies 128-byte read
requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported (KB).
Usage: python tools/pmc_step_summary.py gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write <steps incl. warm-up> <round tag>"""
import csv, glob, json, os, sys
csv.field_size_limit(1 << 30)


def fam(k):
    if 'igemm' in k:
        return 'conv3d igemm, fwd + dgrad'
    if 'wgrad' in k and 'reduce' not in k and 'small' not in k:
        return 'conv3d wgrad'
    return 'everything else'


def load(d, counter):
    out, n = {}, {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = fam(r['Kernel_Name'])
            out[k] = out.get(k, 0.0) + float(r['Counter_Value'])
            n[k] = n.get(k, 0) + 1
    return out, n


fd, wd, steps, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
fetch, nf = load(fd, 'FETCH_SIZE')
write, _ = load(wd, 'WRITE_SIZE')
print(f'# rocprofv3 PMC over the default bench step (config 2, bf16, B=2, 96^3) -- {tag}\n')
print('Collected with `tools/pmc_step.sh` (two separate `--pmc` passes, `--kernel-trace` only), summed per kernel family and divided by the '
      f'{steps} steps of the run.  FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies '
      '128-byte read requests at 64 B).  These counters sit on the L2 -> fabric side, so Infinity-Cache hits are included.\n')
print('| kernel family (launches / step) | FETCH_SIZE raw | FETCH x2 | WRITE_SIZE | total / step |')
print('|---|---|---|---|---|')
conv = 0.0
for k in ('conv3d igemm, fwd + dgrad', 'conv3d wgrad', 'everything else'):
    f, w = fetch.get(k, 0) * 1024 / steps, write.get(k, 0) * 1024 / steps
    print(f'| {k} ({nf.get(k, 0) / steps:.0f}) | {f / 1e9:.2f} GB | {2 * f / 1e9:.2f} GB | {w / 1e9:.2f} GB | {(2 * f + w) / 1e9:.1f} GB |')
    if k != 'everything else':
        conv += 2 * f + w
print(f'\nConv MFMA kernels: {conv / 1e9:.1f} GB per step (algorithmic operand bytes: forward 2.9 + data gradient 3.7 + weight gradient 2.4 = 9.0 GB).')
json.dump({'conv_bytes_per_step': conv, 'source': f'profiles/{tag}_pmc_step.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/pmc_step.sh)'},
          open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'conv_traffic.json'), 'w'))
