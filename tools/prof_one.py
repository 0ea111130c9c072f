#!/usr/bin/env python3
"""Run one conv kernel configuration a few times (for rocprofv3 --pmc / --kernel-trace).  Usage: prof_one.py <layer-index> <fwd|dgrad|wgrad> [xhat]
(xhat: the sources carry no statistics = a pre-normalised input; the weight gradient then runs the LDS-DMA fed kernel)"""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops
LAYERS = [('inc 32->32', 96, 32, 0, 32, False), ('up4.0 96->32+sc', 96, 32, 64, 32, True), ('down1.0 32->64+sc', 48, 32, 0, 64, True),
          ('64->64', 48, 64, 0, 64, False), ('up3.0 192->64+sc', 48, 64, 128, 64, True), ('128->128', 24, 128, 0, 128, False),
          ('up2.0 384->128+sc', 24, 128, 256, 128, True), ('256->256', 12, 256, 0, 256, False), ('320->320', 6, 320, 0, 320, False)]
name, S, Ca, Cb, Cout, sc = LAYERS[int(sys.argv[1])]
which = sys.argv[2]
dt, dev, N = torch.bfloat16, 'cuda', 2
dims = (N, S, S, S); Cin = Ca + Cb
xa = torch.randn((N, S, S, S, Ca), device=dev).to(dt)
xb = torch.randn((N, S, S, S, Cb), device=dev).to(dt) if Cb else None
mra = torch.stack([torch.zeros(N, Ca, device=dev), torch.ones(N, Ca, device=dev)], -1).contiguous()
mrb = torch.stack([torch.zeros(N, Cb, device=dev), torch.ones(N, Cb, device=dev)], -1).contiguous() if Cb else None
w1 = torch.randn((Cout, Cin, 3, 3, 3), device=dev) / math.sqrt(27 * Cin)
ws = torch.randn((Cout, Cin, 3, 3, 3), device=dev) / math.sqrt(27 * Cin) if sc else None
nc = Cout * (2 if sc else 1)
tiles = ops._L().rsuper_conv3_tiles(S, S, S)
if len(sys.argv) > 3 and sys.argv[3] == 'xhat':
    mra = mrb = None
sa, sb = ops.Src(xa, mr=mra), (ops.Src(xb, mr=mrb) if Cb else None)
dy1 = torch.randn((N, S, S, S, Cout), device=dev).to(dt)
dy2 = torch.randn((N, S, S, S, Cout), device=dev).to(dt) if sc else None
if which == 'fwd':
    bn = ops.pick_bn(nc, dt, tiles * N, dims, epi=0); wp = ops.pack_weights(dt, 0, w1, ws, Ca, Cb, Cout, Cout if sc else 0, bn)
    out = torch.empty((N, S, S, S, nc), device=dev, dtype=dt); part = ops.part_buffer(dt, dims, nc, bn, dev)
    fn = lambda: ops.igemm(0, sa, sb, wp, nc, bn, dims, out, part=part)
elif which == 'dgrad':
    bn = ops.pick_bn(Cin, dt, tiles * N, dims, epi=1); wp = ops.pack_weights(dt, 1, w1, ws, Cout, Cout if sc else 0, Cin, 0, bn)
    g0 = torch.empty((N, S, S, S, Cin), device=dev, dtype=dt); part = ops.part_buffer(dt, dims, Cin, bn, dev, epi=1)
    fn = lambda: ops.igemm(1, ops.Src(dy1), ops.Src(dy2) if sc else None, wp, Cin, bn, dims, g0, part=part, ea=sa, eb=sb)
else:
    dw1 = torch.zeros_like(w1); dws = torch.zeros_like(ws) if sc else None
    fn = lambda: ops.wgrad(sa, sb, ops.Src(dy1), ops.Src(dy2) if sc else None, dw1, dws, dims)
for _ in range(int(os.environ.get('PROF_ITERS', '3'))):
    fn()
torch.cuda.synchronize()
