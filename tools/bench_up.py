#!/usr/bin/env python3
"""The InstanceNorm backward tail and the trilinear backward at the UNet's up-block shapes (rsuper_in_bwd_finalize, rsuper_upsample_bwd).
RSUPER_UPSAMPLE_BWD4=0 selects the second-generation gather kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsuper_amd.hip import ops
L = ops._L()


def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for S, C, Ca in ((96, 64, 32), (48, 128, 64), (24, 256, 128), (12, 320, 256)):
    N, I = 2, S // 2
    g0 = torch.randn(N, S, S, S, Ca + C, device='cuda').bfloat16()
    x = torch.randn(N, S, S, S, C, device='cuda').bfloat16()
    mr = torch.rand(N, C, 2, device='cuda') + 0.5
    gm = torch.randn(N, C, 2, device='cuda') * 0.1
    dx = torch.empty(N, I, I, I, C, device='cuda', dtype=torch.bfloat16)
    dyb = [None]

    def fin():
        dyb[0] = ops.in_bwd_finalize(ops.Src(g0, C=C, off=Ca), ops.Src(x, mr=mr), gm, C)

    def plain():
        L.rsuper_upsample_bwd(1, dyb[0].data_ptr(), C, dx.data_ptr(), C, N, I, I, I, S, S, S, C, None)
    t0 = timeit(fin); t1 = timeit(plain)
    mb = N * S ** 3 * C * 2 / 1e6
    print(f'{I}^3 -> {S}^3 x {C} ch: in_bwd_finalize {t0:7.1f} us ({3 * mb / t0:.2f} TB/s) | upsample_bwd {t1:7.1f} us ({(mb + mb / 8) / t1:.2f} TB/s)')
