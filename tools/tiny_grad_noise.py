#!/usr/bin/env python3
"""Per-tensor gradient error of the tiny UNet (f32 kernels) against the fp32 reference fixture and against the float64 CPU restatement:
separates kernel error from the fp32 conditioning noise of the deep pre-activation chain."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
import gpu_checks as gc
from oracle import unet_oracle as uo
T = torch.from_numpy
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'unet_tiny.npz'))
for pool in (True, False):
    pre = '' if pool else 'nopool_'
    shapes = uo.unet_param_shapes(1, 8, len(synth.TINY_CLASSES), pool=pool)
    sd64 = {k: T(v).double().requires_grad_(True) for k, v in synth.fill_state_dict(shapes, 3).items()}
    img = synth.image(1, 48, seed=1234)
    y64 = uo.unet_forward(sd64, T(img).double(), pool=pool)
    go = synth.rng(77).standard_normal(tuple(y64.shape)).astype(np.float32) / y64.numel()
    y64.backward(T(go).double())
    from rsuper_amd.model.dim3.unet import UNet
    net = UNet(1, 8, num_classes=len(synth.TINY_CLASSES), block='BasicBlock', norm='in', pool=pool, compute_dtype='f32')
    net.load_state_dict({k: T(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 3).items()})
    net = net.to('cuda')
    y = net(T(img).to('cuda'))['segmentation']
    y.backward(T(go).to('cuda'))
    torch.cuda.synchronize()
    print(f'# pool={pool}: max |err| / max |grad| per tensor: HIP f32 vs fp32 reference | HIP f32 vs float64 restatement | fp32 reference vs float64')
    worst = [0, 0, 0]
    for k, p in net.named_parameters():
        sc = max(g[f'{pre}g_{k}_summary'][2], 1e-12)
        hip = synth.subsample(p.grad.cpu().numpy(), 4096)[0]
        ref = g[f'{pre}g_{k}_sub']
        f64 = synth.subsample(sd64[k].grad.numpy(), 4096)[0]
        e = [np.abs(hip - ref).max() / sc, np.abs(hip - f64).max() / sc, np.abs(ref - f64).max() / sc]
        worst = [max(a, b) for a, b in zip(worst, e)]
        print(f'{k:40s} {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e}')
    print(f'{"worst":40s} {worst[0]:.2e} | {worst[1]:.2e} | {worst[2]:.2e}')
