#!/usr/bin/env python3
"""Diagnostic: relative L2 difference between the bf16 and f32 (parity-mode) HIP paths on identical weights."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.unet import UNet

def l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()

for base, S, K in [(8, 48, 5), (8, 96, 5), (16, 64, 8), (32, 64, 26)]:
    torch.manual_seed(0)
    nets = {}
    sd = None
    out = {}
    for mode in ('f32', 'bf16'):
        net = UNet(1, base, num_classes=K, compute_dtype=mode)
        if sd is None:
            sd = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        net = net.cuda()
        img = torch.from_numpy(synth.image(1, S, seed=1234)).cuda()
        y = net(img)['segmentation']
        go = torch.from_numpy(synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32)).cuda() / y.numel()
        y.backward(go)
        torch.cuda.synchronize()
        out[mode] = (y.detach().cpu(), {k: p.grad.detach().cpu() for k, p in net.named_parameters()})
    gerr = {k: l2(out['bf16'][1][k], out['f32'][1][k]) for k in out['f32'][1]}
    worst = max(gerr, key=gerr.get)
    print(f'base {base} S {S}: logits L2 {l2(out["bf16"][0], out["f32"][0]):.3e}; grads median {np.median(list(gerr.values())):.3e} '
          f'worst {gerr[worst]:.3e} @ {worst}; inc.conv1 {gerr["inc.conv1.weight"]:.3e} outc {gerr["outc.weight"]:.3e}', flush=True)
