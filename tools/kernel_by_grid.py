#!/usr/bin/env python3
"""Per-(kernel, grid) durations from a rocprofv3 --kernel-trace CSV: which launch shapes of one kernel are slow?
Usage: kernel_by_grid.py trace.csv name-regex [steps]"""
import collections, csv, re, sys
pat = re.compile(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if pat.search(name):
        grid = tuple(int(r[k]) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z'))
        wg = int(r['Workgroup_Size_X'])
        agg[(re.sub(r'^void ', '', name.replace('(anonymous namespace)::', '')).split('(')[0][:60], tuple(g // w for g, w in zip(grid, (wg, 1, 1))))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (name, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{name:60s} blocks {str(grid):22s} {len(d) / steps:6.1f} calls/step  avg {sum(d) / len(d) / 1e3:8.1f} us  total {sum(d) / steps / 1e3:9.1f} us/step')
