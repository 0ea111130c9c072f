#!/usr/bin/env python3
"""csrc/pointwise.hip against torch.mm (f32 exact mode, bf16 mode) + timing against the library GEMM on MedFormer's shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rsuper_amd.hip import ops
torch.manual_seed(0)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
bad = 0
for R, K, N in [(27648, 128, 512), (27648, 512, 128), (27648, 128, 256), (3456, 256, 1024), (3456, 1024, 256), (432, 320, 1280), (221184, 64, 128), (54, 256, 512), (100, 36, 20), (33, 4, 4)]:
    x = torch.randn(R, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    ref = x.double() @ w.double().t() + b.double()
    refd = ref.float() @ w
    for comp, tol in ((torch.float32, 2e-6), (torch.bfloat16, 1.5e-2)):
        y = ops.pointwise_gemm(x, w, b, 0, comp)
        dx = ops.pointwise_gemm(ref.float().contiguous(), w, None, 1, comp)
        e1 = float((y.double() - ref).abs().max() / ref.abs().max())
        e2 = float((dx.double() - (ref @ w.double())).abs().max() / (ref @ w.double()).abs().max())
        ok = e1 < tol and e2 < tol
        bad += not ok
        line = f'R{R} K{K} N{N} {str(comp)[6:]:8s} fwd {e1:.1e} dgrad {e2:.1e} {"ok" if ok else "FAIL"}'
        if R >= 432:
            th = t(lambda: ops.pointwise_gemm(x, w, b, 0, comp)); tl = t(lambda: torch.nn.functional.linear(x, w, b))
            gb = (R * K + R * N) * 4 / 1e9
            line += f' | hip {th:7.1f} us ({gb / th * 1e3:5.2f} TB/s) library {tl:7.1f} us'
        print(line, flush=True)
print('FAILURES', bad)
# weight / bias gradient: HIP slab kernel vs the library formulation of round 2 (bmm over slabs + sum, ones-row GEMM)
for R, K, N in [(27648, 128, 512), (27648, 512, 128), (27648, 128, 256), (3456, 256, 1024), (3456, 1024, 256), (432, 320, 1280), (432, 1280, 320), (221184, 64, 128)]:
    x = torch.randn(R, K, device='cuda'); dy = torch.randn(R, N, device='cuda')
    refw, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    for comp, tol in ((torch.float32, 2e-6), (torch.bfloat16, 1.5e-2)):
        dw, db = ops.pointwise_wgrad(dy, x, True, comp)
        e1 = float((dw.double() - refw).abs().max() / refw.abs().max()); e2 = float((db.double() - refb).abs().max() / refb.abs().max())
        ok = e1 < tol and e2 < 2e-6
        bad += not ok
        th = t(lambda: ops.pointwise_wgrad(dy, x, True, comp))
        slabs = next((s for s in (32, 16, 8, 4) if R % s == 0 and R // s >= 512), 1)
        tl = t(lambda: (torch.bmm(dy.reshape(slabs, R // slabs, -1).transpose(1, 2), x.reshape(slabs, R // slabs, -1)).sum(0), dy.sum(0)))
        gb = (R * K + R * N) * 4 / 1e9
        print(f'wgrad R{R} K{K} N{N} {str(comp)[6:]:8s} dw {e1:.1e} db {e2:.1e} {"ok" if ok else "FAIL"} | hip {th:7.1f} us ({gb / th * 1e3:5.2f} TB/s) library {tl:7.1f} us', flush=True)
print('FAILURES', bad)
