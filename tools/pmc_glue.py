#!/usr/bin/env python3
"""Target of the rocprofv3 --pmc passes for the attention-stage kernels (tools/pmc_glue.sh): one forward + backward of the depthwise
convolution and of the stand-alone InstanceNorm + ReLU on the largest MedFormer shapes at 96^3, B = 2."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.hip import ops
for S, C in ((48, 256), (24, 512)):
    x = torch.randn(2, S, S, S, C, device='cuda').requires_grad_(True)
    w = torch.randn(C, 1, 3, 3, 3, device='cuda').requires_grad_(True)
    go = torch.randn(2, S, S, S, C, device='cuda')
    for _ in range(3):
        torch.autograd.grad(ops.DepthwiseConvFn.apply(x, w), (x, w), go)
        torch.autograd.grad(ops.ChannelNormFn.apply(x, 1e-5, True), (x,), go)
torch.cuda.synchronize()
