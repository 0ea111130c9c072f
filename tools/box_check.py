#!/usr/bin/env python3
"""Parity of the volume-fitted K-split igemm (variants 6 / 7 force it) on the conv cases + its per-layer timing against the default choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gpu_checks as gc

CASES = [(1, (8, 8, 8), 64, 0, 64, False, False), (2, (12, 12, 12), 64, 0, 64, False, True), (1, (24, 24, 24), 128, 0, 128, False, True),
         (2, (8, 24, 32), 32, 0, 32, False, True), (1, (12, 20, 48), 64, 64, 128, True, False), (2, (6, 6, 6), 96, 32, 64, True, False),
         (1, (5, 7, 9), 40, 8, 24, True, False), (2, (13, 3, 11), 72, 0, 96, False, True), (1, (4, 4, 4), 8, 0, 8, False, False),
         (2, (6, 6, 6), 320, 0, 320, False, True), (1, (5, 4, 6), 40, 24, 72, True, False), (2, (3, 6, 2), 64, 0, 40, False, True)]
bad = 0
for v in (6, 7):
    for N, S, Ca, Cb, Co, sc, res in CASES:
        r = gc.with_variant(v, gc.check_conv_fwd, 'bf16', N, S, Ca, Cb, Co, sc, res)
        print(('ok  ' if r['ok'] else 'FAIL'), r['name'], f"{r['err']:.2e}", r['note'], flush=True)
        bad += not r['ok']
        r = gc.with_variant(v, gc.check_conv_bwd, 'bf16', N, S, Ca, Cb, Co, sc)
        print(('ok  ' if r['ok'] else 'FAIL'), r['name'], f"{r['err']:.2e}", r['note'], flush=True)
        bad += not r['ok']
print('FAILURES', bad)
sys.exit(1 if bad else 0)
