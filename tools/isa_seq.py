#!/usr/bin/env python3
"""One letter per instruction of a loop body of a gfx950 listing (hipcc -S --cuda-device-only): M mfma, v valu, L ds_read, W ds_write, G vmem, s salu,
w waitcnt, b branch / barrier, n nop -- shows how finely the non-matrix work is interleaved with the MFMA stream.
usage: python tools/isa_seq.py file.s .LBB1_383"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
lab = sys.argv[2]
i0 = next(i for i, l in enumerate(lines) if l.startswith(lab + ':'))
out = []
for l in lines[i0 + 1:]:
    x = l.strip()
    if not x or x.startswith(';') or x.startswith('.'):
        if re.match(r'^\.LBB', x): out.append('|')
        continue
    c = ('M' if x.startswith('v_mfma') else 'L' if x.startswith('ds_read') else 'W' if x.startswith('ds_write') else 'G' if re.match(r'buffer_|global_|scratch_', x)
         else 'w' if x.startswith('s_waitcnt') else 'n' if x.startswith('s_nop') else 'b' if re.match(r's_c?branch|s_barrier', x) else 'v' if x.startswith('v_') else 's')
    out.append(c)
    m = re.match(r's_c?branch\w*\s+(\S+)', x)
    if m and m.group(1) == lab:
        break
s = ''.join(out)
for k in range(0, len(s), 150):
    print(s[k:k + 150])
