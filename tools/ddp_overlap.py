#!/usr/bin/env python3
"""Gradient exchange of one `bench.py --force-ddp` step (1-rank RCCL group: the GradReducer path the driver's N > 1 runs take) from a rocprofv3
kernel trace: per step, every collective launch (RCCL device kernels) with its queue, start offset inside the backward pass, duration, the compute
kernels it overlaps with on the device timeline, and how much of it lies behind the last compute kernel of the backward (= exposed).
Usage: python tools/ddp_overlap.py <kernel_trace.csv>"""
import csv, re, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:60]
is_coll = lambda r: 'nccl' in r['Kernel_Name'].lower() or 'rccl' in r['Kernel_Name'].lower()
ends = [i for i, r in enumerate(rows) if 'adamw_ema' in r['Kernel_Name']]
colls = [r for r in rows if is_coll(r)]
print(f'# {len(rows)} kernel launches, {len(colls)} collective kernels, queues: { {q: sum(1 for r in rows if r.get("Queue_Id") == q) for q in sorted(set(r.get("Queue_Id", "?") for r in rows))} }')
if len(ends) < 6:
    print('# not enough steps in the trace'); sys.exit(0)
lo, hi = ends[-5] + 1, ends[-3] + 1          # one steady-state step: from after an optimiser launch pair to the next one
step = rows[lo:hi]
t0 = int(step[0]['Start_Timestamp'])
bwd0 = next((int(r['Start_Timestamp']) for r in step if 'head_bwd_data' in r['Kernel_Name']), t0)
comp = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in step if not is_coll(r)]
last_bwd_end = max(e for s, e, k in comp if 'adamw' not in k and 'sqnorm' not in k)
print(f'# step: {len(step)} launches, {(int(step[-1]["End_Timestamp"]) - t0) / 1e3:.0f} us; backward starts at +{(bwd0 - t0) / 1e3:.0f} us, its last compute kernel ends at +{(last_bwd_end - t0) / 1e3:.0f} us')
for r in step:
    if not is_coll(r):
        continue
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ov = [(min(e, ce) - max(s, cs), k) for cs, ce, k in comp if min(e, ce) > max(s, cs)]
    ovt = sum(o for o, _ in ov)
    names = sorted(set(k.split('<')[0].split('(')[0] for _, k in ov))
    exposed = max(0, e - max(s, last_bwd_end))
    print(f'collective q{r.get("Queue_Id", "?")} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  overlapped by compute {ovt / 1e3:7.1f} us ({", ".join(names)[:90]})  behind the backward {exposed / 1e3:6.1f} us   {short(r["Kernel_Name"])}')
