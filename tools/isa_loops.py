#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying basic blocks of a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel, every block (or run of
blocks up to the back edge) holding >= MIN MFMAs -- how many LDS / VMEM / VALU / SALU / wait instructions the wave issues per MFMA.
usage: python tools/isa_loops.py file.s [min_mfma]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
MIN = int(sys.argv[2]) if len(sys.argv) > 2 else 40
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for si, st in enumerate(starts):
    en = starts[si + 1] if si + 1 < len(starts) else len(lines)
    blocks, cur, name = [], [], 'entry'
    for l in lines[st:en]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append((name, cur)); name, cur = m.group(1), []
        else:
            cur.append(l.strip())
    blocks.append((name, cur))
    kname = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', lines[st].split(':')[0])[:40]
    for name, b in blocks:
        ins = [l for l in b if l and not l.startswith(';') and not l.startswith('.')]
        nm = sum(1 for l in ins if l.startswith('v_mfma'))
        if nm < MIN:
            continue
        c = lambda p: sum(1 for l in ins if re.match(p, l))
        tot = len(ins)
        print(f'{kname} {name}: {tot} instr, {nm} mfma ({tot / nm:.2f} per mfma) | ds_read {c("ds_read")} ds_write {c("ds_write")} vmem {c("buffer_|global_|scratch_")} '
              f'valu {c("v_(?!mfma)")} salu {c("s_(?!waitcnt|nop|barrier)")} waitcnt {c("s_waitcnt")} nop {c("s_nop")} lane-spill {c("v_readlane|v_writelane")}')
