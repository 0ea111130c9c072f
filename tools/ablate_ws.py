#!/usr/bin/env python3
"""Ablation timings of the weight-stationary igemm (library built with -DRS_WS_ABLATE).  Usage: RSUPER_WS_ABL=<bits> python tools/ablate_ws.py"""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops
dt, dev, N = torch.bfloat16, 'cuda', 2


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


# forward 32->32 @96^3 (NT=1, single chunk, normalised)
S, C = 96, 32
dims = (N, S, S, S)
x = torch.randn((N, S, S, S, C), device=dev).to(dt)
mr = torch.stack([torch.zeros(N, C, device=dev), torch.ones(N, C, device=dev)], -1).contiguous()
w = torch.randn((C, C, 3, 3, 3), device=dev) / math.sqrt(27 * C)
wp = ops.pack_weights(dt, 0, w, None, C, 0, C, 0, 32)
out = torch.empty_like(x); part = ops.part_buffer(dt, dims, C, 32, dev)
t1 = timeit(lambda: ops.igemm(0, ops.Src(x, mr=mr), None, wp, C, 32, dims, out, part=part))
# dgrad 64->64 @48^3 (NT=2, two chunks, raw)
S, C = 48, 64
dims = (N, S, S, S)
x = torch.randn((N, S, S, S, C), device=dev).to(dt)
dy = torch.randn((N, S, S, S, C), device=dev).to(dt)
mr = torch.stack([torch.zeros(N, C, device=dev), torch.ones(N, C, device=dev)], -1).contiguous()
w = torch.randn((C, C, 3, 3, 3), device=dev) / math.sqrt(27 * C)
wp = ops.pack_weights(dt, 1, w, None, C, 0, C, 0, 32)
g = torch.empty_like(x); part = ops.part_buffer(dt, dims, C, 32, dev, epi=1)
t2 = timeit(lambda: ops.igemm(1, ops.Src(dy), None, wp, C, 32, dims, g, part=part, ea=ops.Src(x, mr=mr)))
print(f'ABL={os.environ.get("RSUPER_WS_ABL", "0"):>2s}  fwd 32->32@96: {t1:7.1f} us   dgrad 64->64@48: {t2:7.1f} us', flush=True)
