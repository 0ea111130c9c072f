#!/usr/bin/env python3
"""Per-tensor gradient errors of the tiny MedFormer (HIP path) against the reference fixture -- debugging aid."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.medformer import MedFormer
T = torch.from_numpy
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'medformer.npz'))
cfg = synth.MEDFORMER_TINY
net = MedFormer(1, len(synth.TINY_CLASSES), compute_dtype=mode, **{k: v for k, v in cfg.items() if k not in ('size', 'seed')})
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: T(v) for k, v in synth.fill_state_dict(shapes, cfg['seed']).items()})
net = net.to('cuda')
y, aux = net(T(synth.image(1, cfg['size'], seed=1234)).to('cuda'))['segmentation']
go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
ga = synth.rng(78).standard_normal(tuple(aux.shape)).astype(np.float32) / aux.numel()
((y * T(go).to('cuda')).sum() + (aux * T(ga).to('cuda')).sum()).backward()
torch.cuda.synchronize()
for k, p in net.named_parameters():
    gsub = synth.subsample(p.grad.cpu().numpy(), 1024)[0]
    ref = g[f'g_{k}_sub']
    sc = max(g[f'g_{k}_summary'][2], 1e-30)
    print(f'{k:70s} err/max {np.abs(gsub - ref).max() / sc:9.2e}  max|ref| {sc:9.2e}  max|got| {np.abs(gsub).max():9.2e}')

# float64 restatement: separates kernel error from the fp32 noise of the reference fixture itself
from oracle import medformer_oracle as mo
sd = {k: T(v).double().requires_grad_(True) for k, v in synth.fill_state_dict(shapes, cfg['seed']).items()}
y64, a64 = mo.medformer_forward(sd, T(synth.image(1, cfg['size'], seed=1234)).double(), cfg)
((y64 * T(go).double()).sum() + (a64 * T(ga).double()).sum()).backward()
gmax = max(float(v.grad.abs().max()) for v in sd.values())
w = [0, 0, 0]
for k, p in net.named_parameters():
    hip = synth.subsample(p.grad.cpu().numpy(), 1024)[0]
    ref = g[f'g_{k}_sub']
    f64 = synth.subsample(sd[k].grad.numpy(), 1024)[0]
    sc = max(np.abs(f64).max(), 1e-3 * gmax)
    e = [np.abs(hip - ref).max() / sc, np.abs(hip - f64).max() / sc, np.abs(ref - f64).max() / sc]
    w = [max(a, b) for a, b in zip(w, e)]
print(f'# worst (scale = max(|g64|, 1e-3 global max {gmax:.2e})): HIP vs reference {w[0]:.2e} | HIP vs float64 {w[1]:.2e} | reference vs float64 {w[2]:.2e}')
