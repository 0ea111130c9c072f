#!/usr/bin/env python3
"""Run one pointwise product (csrc/pointwise.hip) a few times (for rocprofv3 --pmc).  Usage: prof_pw.py <fwd|dgrad|wgrad> [R K N]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops
which = sys.argv[1]
R, K, N = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (27648, 128, 512)
x = torch.randn(R, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; dy = torch.randn(R, N, device='cuda')
fn = {'fwd': lambda: ops.pointwise_gemm(x, w, None, 0, torch.bfloat16), 'dgrad': lambda: ops.pointwise_gemm(dy, w, None, 1, torch.bfloat16),
      'wgrad': lambda: ops.pointwise_wgrad(dy, x, False, torch.bfloat16)}[which]
for _ in range(3):
    fn()
torch.cuda.synchronize()
