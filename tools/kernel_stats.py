#!/usr/bin/env python3
"""Per-kernel table (calls, total, avg, %) from a rocprofv3 --kernel-trace CSV (`*_kernel_trace.csv`), skipping the first
`skip` dispatches (warm-up).  Usage: python tools/kernel_stats.py <kernel_trace.csv> [steps] [skip_fraction]"""
import csv, re, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:96]


agg = {}
for r in rows:
    k = short(r['Kernel_Name'])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(a[1] for a in agg.values())
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print(f'# {len(rows)} dispatches over {steps} steps: {len(rows) / steps:.0f} launches / step, kernel time {tot / 1e6 / steps:.3f} ms / step, '
      f'wall span {span / 1e6:.1f} ms')
print(f'{"kernel":98s} {"calls/step":>10s} {"ms/step":>9s} {"avg_us":>9s} {"pct":>6s}')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:98s} {a[0] / steps:10.1f} {a[1] / 1e6 / steps:9.3f} {a[1] / a[0] / 1e3:9.1f} {100 * a[1] / tot:6.2f}')

# idle time between consecutive dispatches (steady-state half of the trace): where does the queue run dry?
half = rows[len(rows) // 2:]
gaps = {}
idle = 0
for a, b in zip(half, half[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 0:
        idle += g
        k = (short(a['Kernel_Name'])[:44], short(b['Kernel_Name'])[:44])
        e = gaps.setdefault(k, [0, 0])
        e[0] += 1; e[1] += g
hs = steps * len(half) / len(rows)
span_h = int(half[-1]['End_Timestamp']) - int(half[0]['Start_Timestamp'])
print(f'\n# idle between dispatches: {idle / 1e6 / hs:.3f} ms / step of {span_h / 1e6 / hs:.3f} ms / step wall (second half of the trace)')
print(f'{"after":46s} {"before":46s} {"n/step":>7s} {"us/step":>9s} {"avg_us":>8s}')
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{a:46s} {b:46s} {n / hs:7.1f} {t / 1e3 / hs:9.1f} {t / n / 1e3:8.1f}')
