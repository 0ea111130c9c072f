#!/usr/bin/env python3
"""The launch sequence of ONE steady-state step from a rocprofv3 --kernel-trace CSV: index, stream, start offset, duration, gap to the previous kernel on
the device timeline, kernel name + grid -- to see which small launches sit next to each other (fusion candidates) and where the queue runs dry.
Usage: python tools/kernel_seq.py <kernel_trace.csv> <steps in the trace> [step index, default: the second to last]"""
import csv, re, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
steps = int(sys.argv[2])
per = len(rows) // steps
k = int(sys.argv[3]) if len(sys.argv) > 3 else steps - 2
# align on the optimiser kernel that ends a step
ends = [i for i, r in enumerate(rows) if 'adamw_ema' in r['Kernel_Name']]
if len(ends) >= 4:
    lo, hi = ends[-4] + 1, ends[-2] + 1
else:
    lo, hi = k * per, (k + 1) * per
sel = rows[lo:hi]
t0 = int(sel[0]['Start_Timestamp'])
prev_end = t0
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'at::native::', '', n)
    return n[:70]
print(f'# {len(sel)} launches, {(int(sel[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us')
for i, r in enumerate(sel):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f'{i:4d} q{r.get("Queue_Id", "?"):>3s} +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {short(r["Kernel_Name"])}  grid {r.get("Grid_Size", "")}/{r.get("Workgroup_Size", "")}')
    prev_end = max(prev_end, e)
