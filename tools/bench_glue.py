#!/usr/bin/env python3
"""Timings of the MedFormer attention-stage kernels (depthwise 3x3x3, stand-alone InstanceNorm) on the shapes of the shipped
configuration at 96^3, B = 2, against their HBM floors.  Usage: python tools/bench_glue.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print('| op | shape (N,D,H,W,C) | MB | fwd us | GB/s | bwd us (dx+dw) | GB/s |')
print('|---|---|---|---|---|---|---|')
for S, C in ((48, 256), (24, 512), (24, 128), (24, 384), (12, 1024), (12, 256), (6, 1280)):
    x = torch.randn(2, S, S, S, C, device='cuda').requires_grad_(True)
    w = torch.randn(C, 1, 3, 3, 3, device='cuda').requires_grad_(True)
    mb = x.numel() * 4 / 1e6
    y = ops.DepthwiseConvFn.apply(x, w)
    go = torch.randn_like(y)
    tf = timed(lambda: ops.DepthwiseConvFn.apply(x, w))
    tb = timed(lambda: torch.autograd.grad(ops.DepthwiseConvFn.apply(x, w), (x, w), go)) - tf
    print(f'| depthwise3 | (2,{S},{S},{S},{C}) | {mb:.0f} | {tf:.1f} | {2 * mb / tf * 1e3:.0f} | {tb:.1f} | {4 * mb / tb * 1e3:.0f} |')
    yn = ops.ChannelNormFn.apply(x, 1e-5, True)
    tf = timed(lambda: ops.ChannelNormFn.apply(x, 1e-5, True))
    tb = timed(lambda: torch.autograd.grad(ops.ChannelNormFn.apply(x, 1e-5, True), (x,), go)) - tf
    print(f'| instnorm+relu | (2,{S},{S},{S},{C}) | {mb:.0f} | {tf:.1f} | {3 * mb / tf * 1e3:.0f} | {tb:.1f} | {5 * mb / tb * 1e3:.0f} |')
