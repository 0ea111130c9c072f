#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the layer shapes of the full UNet (B=2, 96^3, base 32).
Usage: python tools/bench_conv.py [bf16|f32] [variant|-] [batch]"""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
if len(sys.argv) > 2 and sys.argv[2] != '-':
    ops._L().rsuper_conv3_variant(int(sys.argv[2]))      # force an igemm kernel variant
dt = {'bf16': torch.bfloat16, 'f32': torch.float32}[mode]
dev = 'cuda'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # per-GPU batch (the layer times at a larger batch show the under-fill of the low levels)
# (name, S, Ca, Cb, Cout, fused_sc)
LAYERS = [('inc 32->32', 96, 32, 0, 32, False), ('up4.0 96->32+sc', 96, 32, 64, 32, True), ('down1.0 32->64+sc', 48, 32, 0, 64, True),
          ('64->64', 48, 64, 0, 64, False), ('up3.0 192->64+sc', 48, 64, 128, 64, True), ('128->128', 24, 128, 0, 128, False),
          ('up2.0 384->128+sc', 24, 128, 256, 128, True), ('256->256', 12, 256, 0, 256, False), ('320->320', 6, 320, 0, 320, False),
          # the fused [conv1 | shortcut] members of the low-resolution blocks (round 6: up1.0's weight gradient was the slowest per FLOP of the step and in no table)
          ('down2.0 64->128+sc', 24, 64, 0, 128, True), ('down3.0 128->256+sc', 12, 128, 0, 256, True), ('down4.0 256->320+sc', 6, 256, 0, 320, True),
          ('up1.0 576->256+sc', 12, 256, 320, 256, True)]

if os.environ.get('BC_EXTRA'):                           # extra rows: the two column ranges of up4.0's data gradient as launches of their own (K = 64 -> 64 / 32 columns)
    LAYERS = [('up4.0 dgrad cols 32..95', 96, 64, 0, 32, True), ('up4.0 dgrad cols 0..31', 96, 32, 0, 32, True)]
if os.environ.get('BC_LAYERS'):                        # custom rows: "name:S:Ca:Cb:Cout:sc;..." (sc = 0 / 1)
    LAYERS = [(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5] == '1') for f in (r.split(':') for r in os.environ['BC_LAYERS'].split(';'))]
if os.environ.get('BC_ONLY_S2'):                       # only the strided rows below
    LAYERS = []
if os.environ.get('BC_ONLY'):                          # only the stride-1 layers whose name contains one of the comma-separated substrings
    LAYERS = [l for l in LAYERS if any(k in l[0] for k in os.environ['BC_ONLY'].split(','))]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, S, Ca, Cb, Cout, sc in LAYERS:
    dims = (N, S, S, S)
    Cin = Ca + Cb
    xa = torch.randn((N, S, S, S, Ca), device=dev).to(dt)
    xb = torch.randn((N, S, S, S, Cb), device=dev).to(dt) if Cb else None
    mra = torch.stack([torch.zeros(N, Ca, device=dev), torch.ones(N, Ca, device=dev)], -1).contiguous()
    mrb = torch.stack([torch.zeros(N, Cb, device=dev), torch.ones(N, Cb, device=dev)], -1).contiguous() if Cb else None
    w1 = torch.randn((Cout, Cin, 3, 3, 3), device=dev) / math.sqrt(27 * Cin)
    ws = torch.randn((Cout, Cin, 3, 3, 3), device=dev) / math.sqrt(27 * Cin) if sc else None
    nc = Cout * (2 if sc else 1)
    tiles = ops._L().rsuper_conv3_tiles(S, S, S)
    sa, sb = ops.Src(xa, mr=mra), (ops.Src(xb, mr=mrb) if Cb else None)
    # forward
    bn = ops.pick_bn(nc, dt, tiles * N, dims, epi=0)
    wp = ops.pack_weights(dt, 0, w1, ws, Ca, Cb, Cout, Cout if sc else 0, bn)
    out = torch.empty((N, S, S, S, nc), device=dev, dtype=dt)
    part = ops.part_buffer(dt, (N, S, S, S), nc, bn, dev)
    t_f = timeit(lambda: ops.igemm(0, sa, sb, wp, nc, bn, dims, out, part=part))
    fl = 2.0 * N * S ** 3 * nc * Cin * 27
    # dgrad
    dy1 = torch.randn((N, S, S, S, Cout), device=dev).to(dt)
    dy2 = torch.randn((N, S, S, S, Cout), device=dev).to(dt) if sc else None
    bnd = ops.pick_bn(Cin, dt, tiles * N, dims, epi=1)
    wpd = ops.pack_weights(dt, 1, w1, ws, Cout, Cout if sc else 0, Cin, 0, bnd)
    g0 = torch.empty((N, S, S, S, Cin), device=dev, dtype=dt)
    partd = ops.part_buffer(dt, (N, S, S, S), Cin, bnd, dev, epi=1)
    t_d = timeit(lambda: ops.igemm(1, ops.Src(dy1), ops.Src(dy2) if sc else None, wpd, Cin, bnd, dims, g0, part=partd, ea=sa, eb=sb))
    # wgrad
    dw1 = torch.zeros_like(w1); dws = torch.zeros_like(ws) if sc else None
    t_w = timeit(lambda: ops.wgrad(sa, sb, ops.Src(dy1), ops.Src(dy2) if sc else None, dw1, dws, dims))
    print(f'{name:20s} S{S:3d} {fl / 1e9:7.1f} GF | fwd bn{bn:3d} {t_f * 1e3:8.1f} us {fl / t_f / 1e9:7.1f} TF | dgrad bn{bnd:3d} {t_d * 1e3:8.1f} us {fl / t_d / 1e9:7.1f} TF | '
          f'wgrad {t_w * 1e3:8.1f} us {fl / t_w / 1e9:7.1f} TF', flush=True)

# ---- the strided member (down_block(pool=False)): [conv1 | shortcut] of BasicBlock(Cin, Cout, stride=2) on the parity-class kernel against the
#      rounds-1/2 evaluation (the stride-1 GEMM at full resolution + rsuper_subsample2 / zero-stuffed dy).  FLOPs = the strided convolution's own.
print('# stride-2 [conv1 | shortcut]: parity-class kernel (conv3d_igemm_s2.hip) vs stride-1 evaluation at full resolution', flush=True)
for name, S, Ca, Cout in ([] if (os.environ.get('BC_ONLY') or os.environ.get('BC_LAYERS')) else [('down1.0 s2 32->64+sc', 96, 32, 64), ('down2.0 s2 64->128+sc', 48, 64, 128), ('down3.0 s2 128->256+sc', 24, 128, 256)]):
    dims = (N, S, S, S); O = (S + 1) // 2
    xa = torch.randn((N, S, S, S, Ca), device=dev).to(dt)
    mra = torch.stack([torch.zeros(N, Ca, device=dev), torch.ones(N, Ca, device=dev)], -1).contiguous()
    w1 = torch.randn((Cout, Ca, 3, 3, 3), device=dev) / math.sqrt(27 * Ca); ws = torch.randn((Cout, Ca, 3, 3, 3), device=dev) / math.sqrt(27 * Ca)
    nc = 2 * Cout
    sa = ops.Src(xa, mr=mra)
    wp = ops.pack_weights(dt, 0, w1, ws, Ca, 0, Cout, Cout, 64)
    ys = torch.empty((N, O, O, O, nc), device=dev, dtype=dt)
    part = torch.empty((N, ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 1, Ca, 0, nc, N, S, S, S), nc, 2), device=dev, dtype=torch.float32)
    t_f = timeit(lambda: ops.igemm_s2(1, sa, None, wp, nc, dims, ys, part))
    tiles = ops._L().rsuper_conv3_tiles(S, S, S)
    bn = ops.pick_bn(nc, dt, tiles * N, dims)
    wpl = ops.pack_weights(dt, 0, w1, ws, Ca, 0, Cout, Cout, bn)
    full = torch.empty((N, S, S, S, nc), device=dev, dtype=dt)
    t_fl = timeit(lambda: (ops.igemm(0, sa, None, wpl, nc, bn, dims, full), ops.subsample2(full)))
    dy1 = torch.randn((N, O, O, O, Cout), device=dev).to(dt); dy2 = torch.randn((N, O, O, O, Cout), device=dev).to(dt)
    wpd = ops.pack_weights(dt, 1, w1, ws, Cout, Cout, Ca, 0, 64)
    g0 = torch.empty((N, S, S, S, Ca), device=dev, dtype=dt)
    partd = torch.empty((N, ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 2, Cout, Cout, Ca, N, S, S, S), Ca, 2), device=dev, dtype=torch.float32)
    t_d = timeit(lambda: ops.igemm_s2(2, ops.Src(dy1), ops.Src(dy2), wpd, Ca, dims, g0, partd, ea=sa))
    bnd = ops.pick_bn(Ca, dt, tiles * N, dims)
    wpdl = ops.pack_weights(dt, 1, w1, ws, Cout, Cout, Ca, 0, bnd)
    dfull = torch.empty((N, S, S, S, nc), device=dev, dtype=dt)
    partl = ops.part_buffer(dt, dims, Ca, bnd, dev, epi=1)
    t_dl = timeit(lambda: (ops.subsample2_scatter(dy1, dfull, 0, dims), ops.subsample2_scatter(dy2, dfull, Cout, dims),
                           ops.igemm(1, ops.Src(dfull, C=Cout), ops.Src(dfull, C=Cout, off=Cout), wpdl, Ca, bnd, dims, g0, part=partl, ea=sa)))
    dw1 = torch.zeros_like(w1); dws = torch.zeros_like(ws)
    t_w = timeit(lambda: ops.wgrad_s2(sa, ops.Src(dy1), ops.Src(dy2), dw1, dws, dims))
    t_wl = timeit(lambda: ops.wgrad(sa, None, ops.Src(dfull, C=Cout), ops.Src(dfull, C=Cout, off=Cout), dw1, dws, dims))   # + the stuffing pass timed with the dgrad
    fl = 2.0 * N * O ** 3 * nc * Ca * 27
    print(f'{name:22s} S{S:3d}->{O:2d} {fl / 1e9:6.1f} GF | fwd {t_f * 1e3:7.1f} us {fl / t_f / 1e9:6.1f} TF (full-res + subsample {t_fl * 1e3:7.1f} us: x{t_fl / t_f:4.1f}) | '
          f'dgrad {t_d * 1e3:7.1f} us {fl / t_d / 1e9:6.1f} TF (zero-stuffed full-res {t_dl * 1e3:7.1f} us: x{t_dl / t_d:4.1f}) | '
          f'wgrad {t_w * 1e3:7.1f} us {fl / t_w / 1e9:6.1f} TF (stride-1 kernel on the zero-stuffed dy {t_wl * 1e3:7.1f} us: x{t_wl / t_w:4.1f})', flush=True)
