#!/usr/bin/env python3
"""what would a pre-normalised bf16 copy x_hat of a layer input buy?  The conv kernels skip the InstanceNorm + ReLU
VALU of their staging when a source carries no statistics (mr = None), so the SAME kernels are timed on a raw source (norm + ReLU while
staging, as shipped) and on a source without statistics (= an x_hat tensor materialised beforehand), next to the cost of one
read-2-bytes / write-2-bytes pass over the tensor (what the materialisation itself would cost per layer input).
Usage: python tools/xhat_probe.py"""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops

dt, dev, N = torch.bfloat16, 'cuda', 2
LAYERS = [('32->32', 96, 32, 0, 32, False), ('up4.0 96->32+sc', 96, 32, 64, 32, True), ('down1.0 32->64+sc', 48, 32, 0, 64, True),
          ('64->64', 48, 64, 0, 64, False), ('up3.0 192->64+sc', 48, 64, 128, 64, True), ('128->128', 24, 128, 0, 128, False)]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


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
    bn = ops.pick_bn(nc, dt, tiles * N, dims)
    wp = ops.pack_weights(dt, 0, w1, ws, Ca, Cb, Cout, Cout if sc else 0, bn)
    out = torch.empty((N, S, S, S, nc), device=dev, dtype=dt)
    part = ops.part_buffer(dt, dims, nc, bn, dev)
    dy1 = torch.randn((N, S, S, S, Cout), device=dev).to(dt)
    dy2 = torch.randn((N, S, S, S, Cout), device=dev).to(dt) if sc else None
    dw1 = torch.zeros_like(w1); dws = torch.zeros_like(ws) if sc else None
    res = {}
    for tag, (ma, mb) in (('raw', (mra, mrb)), ('xhat', (None, None))):
        sa, sb = ops.Src(xa, mr=ma), (ops.Src(xb, mr=mb) if Cb else None)
        res[tag] = (timeit(lambda: ops.igemm(0, sa, sb, wp, nc, bn, dims, out, part=part)),
                    timeit(lambda: ops.wgrad(sa, sb, ops.Src(dy1), ops.Src(dy2) if sc else None, dw1, dws, dims)))
    cost = timeit(lambda: torch.clamp_min(xa, 0)) + (timeit(lambda: torch.clamp_min(xb, 0)) if Cb else 0.0)
    fl = 2.0 * N * S ** 3 * nc * Cin * 27
    print(f'{name:20s} S{S:3d} | fwd raw {res["raw"][0]:7.1f} us  x_hat {res["xhat"][0]:7.1f} us | wgrad raw {res["raw"][1]:7.1f} us  x_hat {res["xhat"][1]:7.1f} us | '
          f'saved {res["raw"][0] - res["xhat"][0] + res["raw"][1] - res["xhat"][1]:6.1f} us vs one 2B-read / 2B-write pass over the input {cost:6.1f} us', flush=True)
