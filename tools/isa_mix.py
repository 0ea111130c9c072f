#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 assembly dump (hipcc -S --cuda-device-only).  Usage: isa_mix.py file.s [name-regex]"""
import re, sys
s = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
parts = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', s)
for i in range(1, len(parts) - 1, 2):
    name, body = parts[i], parts[i + 1]
    if pat and not pat.search(name):
        continue
    body = body.split('.section')[0]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    def cnt(p):
        return sum(1 for l in lines if re.match(p, l))
    print(name[:100])
    print('   mfma', cnt(r'v_mfma'), 'accread', cnt(r'v_accvgpr_read'), 'accwrite', cnt(r'v_accvgpr_write'), 'ds_read', cnt(r'ds_read'), 'ds_write', cnt(r'ds_write'),
          'buffer_load', cnt(r'buffer_load'), 'buffer_store', cnt('buffer_store'), 'scratch', cnt(r'scratch_'), 'valu', cnt(r'v_(?!mfma|accvgpr)'), 'salu', cnt(r's_(?!waitcnt|nop|barrier)'),
          'waitcnt', cnt(r's_waitcnt'), 'nop', cnt('s_nop'), 'barrier', cnt('s_barrier'), 'total', len(lines))
