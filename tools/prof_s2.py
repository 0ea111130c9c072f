#!/usr/bin/env python3
"""Run the strided (parity-class) igemm / the strided weight gradient a few times (for rocprofv3 --pmc).  Usage: prof_s2.py <fwd|dgrad|wgrad> [S Ca Cout]"""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops
which = sys.argv[1]
S, Ca, Cout = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (96, 32, 64)
dt, dev, N = torch.bfloat16, 'cuda', 2
dims = (N, S, S, S); O = (S + 1) // 2; nc = 2 * Cout
xa = torch.randn((N, S, S, S, Ca), device=dev).to(dt)
mra = torch.stack([torch.zeros(N, Ca, device=dev), torch.ones(N, Ca, device=dev)], -1).contiguous()
w1 = torch.randn((Cout, Ca, 3, 3, 3), device=dev) / math.sqrt(27 * Ca); ws = torch.randn((Cout, Ca, 3, 3, 3), device=dev) / math.sqrt(27 * Ca)
sa = ops.Src(xa, mr=mra)
if which == 'fwd':
    wp = ops.pack_weights(dt, 0, w1, ws, Ca, 0, Cout, Cout, 64)
    ys = torch.empty((N, O, O, O, nc), device=dev, dtype=dt)
    part = torch.empty((N, ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 1, Ca, 0, nc, N, S, S, S), nc, 2), device=dev, dtype=torch.float32)
    fn = lambda: ops.igemm_s2(1, sa, None, wp, nc, dims, ys, part)
elif which == 'wgrad':
    dy1 = torch.randn((N, O, O, O, Cout), device=dev).to(dt); dy2 = torch.randn((N, O, O, O, Cout), device=dev).to(dt)
    dw1 = torch.zeros_like(w1); dws = torch.zeros_like(ws)
    fn = lambda: ops.wgrad_s2(sa, ops.Src(dy1), ops.Src(dy2), dw1, dws, dims)
else:
    dy1 = torch.randn((N, O, O, O, Cout), device=dev).to(dt); dy2 = torch.randn((N, O, O, O, Cout), device=dev).to(dt)
    wpd = ops.pack_weights(dt, 1, w1, ws, Cout, Cout, Ca, 0, 64)
    g0 = torch.empty((N, S, S, S, Ca), device=dev, dtype=dt)
    partd = torch.empty((N, ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 2, Cout, Cout, Ca, N, S, S, S), Ca, 2), device=dev, dtype=torch.float32)
    fn = lambda: ops.igemm_s2(2, ops.Src(dy1), ops.Src(dy2), wpd, Ca, dims, g0, partd, ea=sa)
for _ in range(3):
    fn()
torch.cuda.synchronize()
