#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (.db) into a per-kernel table (calls, total, avg, %),
like `--stats` prints.  Usage: python profiles/summarize_rocpd.py <results.db> [skip_first_n_dispatches]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute('select s.kernel_name, d.start, d.end, d.grid_size_x*d.grid_size_y*d.grid_size_z/ (d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z), d.group_segment_size, s.arch_vgpr_count '
                      'from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start').fetchall()
    agg = {}
    for name, st, en, wgs, lds, vg in rows:
        a = agg.setdefault(short(name), [0, 0, 0, lds, vg])
        a[0] += 1
        a[1] += en - st
        a[2] = max(a[2], wgs)
    total = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f'# {len(rows)} dispatches, kernel time {total / 1e6:.3f} ms, wall span {span / 1e6:.3f} ms (GPU busy {100 * total / span:.1f}%)')
    print(f'{"kernel":112s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"pct":>6s} {"maxWGs":>7s} {"LDS":>7s} {"VGPR":>5s}')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:112s} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.1f} {100 * a[1] / total:6.2f} {a[2]:7d} {a[3]:7d} {a[4]:5d}')


if __name__ == '__main__':
    main()
