#!/usr/bin/env python3
"""Stem / head kernels at the benchmark shape (B = 2, 96^3, base 32, 26 classes): forward and backward launches in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsuper_amd.hip import ops


def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


N, S, C, K = 2, 96, 32, 26
dt = torch.bfloat16
img = torch.randn(N, 1, S, S, S, device='cuda')
w = (torch.randn(C, 1, 3, 3, 3, device='cuda') * 0.2).requires_grad_(True)
y, mr = ops.StemFn.apply(img, w, dt)
go = torch.randn_like(y)
print(f'stem fwd  {timeit(lambda: ops.StemFn.apply(img, w, dt)):7.1f} us (includes stats_finalize)')
def sb():
    w.grad = None
    yy, _ = ops.StemFn.apply(img, w, dt)
    yy.backward(go)
t_fb = timeit(sb)
print(f'stem fwd + wgrad {t_fb:7.1f} us')
x = torch.randn(N, S, S, S, C, device='cuda').to(dt).requires_grad_(True)
hw = (torch.randn(K, C, 1, 1, 1, device='cuda') * 0.3).requires_grad_(True)
hb = torch.zeros(K, device='cuda', requires_grad=True)
print(f'head fwd  {timeit(lambda: ops.HeadFn.apply(x, hw, hb)):7.1f} us')
lg = ops.HeadFn.apply(x, hw, hb)
gl = torch.randn_like(lg)
def hbk():
    x.grad = None; hw.grad = None; hb.grad = None
    ops.HeadFn.apply(x, hw, hb).backward(gl)
print(f'head fwd + bwd {timeit(hbk):7.1f} us')
