#!/usr/bin/env python3
"""Host cost per call of the custom autograd ops on tiny tensors (the GPU work is negligible: what remains is Python + ctypes + autograd
bookkeeping) -- the budget of the eager, host-bound MedFormer step.  Usage: python tools/op_host_cost.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops
from rsuper_amd.model.dim3 import medformer_utils as mu


def cost(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


dev = 'cuda'
x = torch.randn(2, 4, 4, 4, 64, device=dev, requires_grad=True)
xb = torch.randn(2, 12, 12, 12, 64, device=dev, requires_grad=True)
w = torch.randn(64, 1, 3, 3, 3, device=dev, requires_grad=True)
wl = torch.randn(64, 64, device=dev, requires_grad=True)
fqv = torch.randn(2, 64, 128, device=dev, requires_grad=True)
mqv = torch.randn(2, 27, 128, device=dev, requires_grad=True)
se = [torch.randn(16, 64, 1, 1, 1, device=dev, requires_grad=True), torch.randn(16, device=dev, requires_grad=True),
      torch.randn(64, 16, 1, 1, 1, device=dev, requires_grad=True), torch.randn(64, device=dev, requires_grad=True)]


def fb(f):
    def run():
        y = f()
        y = y[0] if isinstance(y, tuple) else y
        y.backward(torch.ones_like(y))
    return run


rows = [('torch.empty((2, 64, 2))', lambda: torch.empty((2, 64, 2), device=dev)),
        ('x + x (ATen elementwise)', lambda: x.detach() + x.detach()),
        ('ChannelNormFn fwd (one-launch path)', lambda: ops.ChannelNormFn.apply(x.detach(), 1e-5, True)),
        ('ChannelNormFn fwd (three-launch path)', lambda: ops.ChannelNormFn.apply(xb.detach(), 1e-5, True)),
        ('ChannelNormFn fwd + bwd', fb(lambda: ops.ChannelNormFn.apply(xb, 1e-5, True))),
        ('DepthwiseConvFn fwd', lambda: ops.DepthwiseConvFn.apply(x.detach(), w.detach())),
        ('DepthwiseConvFn fwd + bwd', fb(lambda: ops.DepthwiseConvFn.apply(x, w))),
        ('F.linear fwd (library GEMM)', lambda: torch.nn.functional.linear(x.detach(), wl.detach())),
        ('F.linear fwd + bwd', fb(lambda: torch.nn.functional.linear(x, wl))),
        ('BidirAttnFn fwd', lambda: ops.BidirAttnFn.apply(fqv.detach(), mqv.detach(), 2, 0.17)),
        ('BidirAttnFn fwd + bwd', fb(lambda: ops.BidirAttnFn.apply(fqv, mqv, 2, 0.17))),
        ('SqueezeExciteFn fwd', lambda: ops.SqueezeExciteFn.apply(x.detach(), *[t.detach() for t in se])),
        ('SqueezeExciteFn fwd + bwd', fb(lambda: ops.SqueezeExciteFn.apply(x, *se)))]
torch.backends.cuda.preferred_blas_library('cublas')
for name, fn in rows:
    print(f'{name:42s} {cost(fn):7.1f} us of host time per call')
