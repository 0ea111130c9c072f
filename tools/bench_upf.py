#!/usr/bin/env python3
"""Trilinear forward (+ statistics) at the UNet's four up-block shapes."""
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


for S, C in ((96, 64), (48, 128), (24, 256), (12, 320)):
    x = torch.randn(2, S // 2, S // 2, S // 2, C, device='cuda').bfloat16()
    t = timeit(lambda: ops.UpsampleFn.apply(x, (S, S, S)))
    mb = 2 * S ** 3 * C * 2 / 1e6 * (1 + 1 / 8)
    print(f'{S // 2}^3 -> {S}^3 x {C} ch: upsample forward + finalize {t:7.1f} us ({mb / t:.2f} TB/s)')
