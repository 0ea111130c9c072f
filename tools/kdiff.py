#!/usr/bin/env python3
"""Diff two tools/kernel_stats.py tables (same box, two builds): per-kernel ms/step before -> after.  Usage: kdiff.py A.txt B.txt"""
import sys


def load(p):
    d = {}
    for ln in open(p):
        if ln.startswith('#') or ln.startswith('kernel') or not ln.strip():
            if ln.startswith('# idle'):
                break
            continue
        name, rest = ln[:98].strip(), ln[98:].split()
        if len(rest) == 4:
            d[name] = (float(rest[0]), float(rest[1]))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
print(f'{"kernel":70s} {"calls A":>8s} {"calls B":>8s} {"ms A":>8s} {"ms B":>8s} {"delta":>8s}')
for k in rows[:40]:
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    print(f'{k[:70]:70s} {ca:8.1f} {cb:8.1f} {ta:8.3f} {tb:8.3f} {tb - ta:+8.3f}')
print(f'{"total":70s} {sum(v[0] for v in a.values()):8.1f} {sum(v[0] for v in b.values()):8.1f} {sum(v[1] for v in a.values()):8.3f} '
      f'{sum(v[1] for v in b.values()):8.3f} {sum(v[1] for v in b.values()) - sum(v[1] for v in a.values()):+8.3f}')
