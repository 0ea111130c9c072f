#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying LOOPS of a gfx950 assembly listing (hipcc -S --cuda-device-only): for every backward branch whose body
(the blocks from the target label to the branch) holds >= MIN MFMAs, how many LDS / VMEM / VALU / SALU / wait instructions a wave issues per MFMA.
Nested loops are reported innermost-first; a body that contains another reported loop is skipped.
usage: python tools/isa_loops.py file.s [min_mfma]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
MIN = int(sys.argv[2]) if len(sys.argv) > 2 else 40
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for si, st in enumerate(starts):
    en = starts[si + 1] if si + 1 < len(starts) else len(lines)
    body = lines[st:en]
    label_at = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    kname = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', lines[st].split(':')[0])[:44]
    seen = []
    loops = []
    for i, l in enumerate(body):
        m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i, m.group(1)))
    loops.sort(key=lambda t: t[1] - t[0])
    for lo, hi, lab in loops:
        if any(lo <= a and b <= hi for a, b in seen):
            continue
        ins = [x.strip() for x in body[lo:hi + 1]]
        ins = [x for x in ins if x and not x.startswith(';') and not x.startswith('.')]
        nm = sum(1 for x in ins if x.startswith('v_mfma'))
        if nm < MIN:
            continue
        seen.append((lo, hi))
        c = lambda p: sum(1 for x in ins if re.match(p, x))
        tot = len(ins)
        print(f'{kname} loop {lab}: {tot} instr, {nm} mfma ({tot / nm:.2f} per mfma) | ds_read {c("ds_read")} ds_write {c("ds_write")} vmem {c("buffer_|global_|scratch_")} '
              f'valu {c("v_(?!mfma)")} salu {c("s_(?!waitcnt|nop|barrier|c?branch)")} branch {c("s_c?branch")} waitcnt {c("s_waitcnt")} nop {c("s_nop")} lane-spill {c("v_readlane|v_writelane")}')
