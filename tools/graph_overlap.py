#!/usr/bin/env python3
"""Do kernels of a replayed step overlap in time?  A hipGraph captured from ONE stream is a chain; overlapping kernel intervals in a rocprofv3
kernel trace show parallel branches (another stream joined the capture) and which kernels sit on them.  Usage: graph_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
pairs = collections.Counter(); n = 0; ovl = collections.defaultdict(list); when = collections.defaultdict(list)
active = []
t0 = rows[0][0]
for s, e, k, q, st in rows:
    active = [a for a in active if a[1] > s]
    for a in active:
        pairs[(a[2], k)] += 1; n += 1
        ovl[(a[2], k)].append(min(a[1], e) - s); when[(a[2], k)].append((s - t0) / 1e6)
    active.append((s, e, k))
print(f'{len(rows)} kernels, {n} overlapping pairs; queues {collections.Counter(r[3] for r in rows).most_common(6)}; streams {collections.Counter(r[4] for r in rows).most_common(6)}')
for (a, b), c in pairs.most_common(25):
    o = sorted(ovl[(a, b)])
    print(f'{c:6d}  overlap ns median {o[len(o) // 2]} max {o[-1]}  at ms {min(when[(a, b)]):.0f}..{max(when[(a, b)]):.0f}  {a[:48]}  ||  {b[:48]}')
print('trace spans ms', (rows[-1][1] - t0) / 1e6)
