#!/usr/bin/env python3
"""Run the registered checks whose function names contain argv[1] (default: conv_exact); optional argv[2] = only under this forced variant (e.g. v8).
Prints one line per check."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import gpu_checks as gc
pat = sys.argv[1] if len(sys.argv) > 1 else 'conv_exact'
only = sys.argv[2] if len(sys.argv) > 2 else None
bad = 0
for fn, a in gc.all_checks():
    names = [fn.__name__] + [x.__name__ for x in a if callable(x)]
    if not any(pat in n for n in names):
        continue
    if only is not None and not (fn is gc.with_variant and f'v{a[0]}' == only):
        continue
    try:
        r = fn(*a)
    except Exception as e:      # keep going: one line per check
        r = dict(ok=False, name=f'{names} {a[:1]}', err=float('nan'), note=f'EXCEPTION {type(e).__name__}: {e}')
    bad += not r['ok']
    print('ok  ' if r['ok'] else 'FAIL', r['name'], f"err {r['err']:.3e}", r['note'], flush=True)
print('failed:', bad)
