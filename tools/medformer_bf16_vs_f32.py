import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch, synth
from rsuper_amd.model.dim3.medformer import MedFormer
cfg = dict(base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
           num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True)
ncls = 5
init = sys.argv[1] if len(sys.argv) > 1 else 'default'
for S in (64, 96):
    out = {}
    for mode in ('f32', 'bf16'):
        torch.manual_seed(0)
        net = MedFormer(1, ncls, compute_dtype=mode, **cfg)
        if init == 'synth':
            shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
            sd_np = synth.fill_state_dict(shapes, 23)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        net = net.to('cuda')
        x = torch.from_numpy(synth.image(1, S, seed=77)).to('cuda')
        go = torch.from_numpy(synth.rng(5).standard_normal((1, ncls, S, S, S)).astype(np.float32)).to('cuda') / S ** 3
        ga = torch.from_numpy(synth.rng(6).standard_normal((1, ncls, S, S, S)).astype(np.float32)).to('cuda') / S ** 3
        y, a = net(x)['segmentation']
        ((y * go).sum() + (a * ga).sum()).backward(); torch.cuda.synchronize()
        out[mode] = (y.detach().double(), a.detach().double(), {k: p.grad.double() for k, p in net.named_parameters()})
    f, b = out['f32'], out['bf16']
    l2 = lambda u, v: float((u - v).norm() / v.norm())
    cos = {k: float((b[2][k] * f[2][k]).sum() / (b[2][k].norm() * f[2][k].norm()).clamp_min(1e-300)) for k in f[2]}
    rl = {k: l2(b[2][k], f[2][k]) for k in f[2]}
    ks = sorted(cos, key=cos.get)
    allg = lambda t: torch.cat([v.flatten() for v in t[2].values()])
    print(S, 'logits relL2', l2(b[0], f[0]), 'aux', l2(b[1], f[1]), 'all-grads relL2', l2(allg(b), allg(f)), 'cos all', float((allg(b) * allg(f)).sum() / (allg(b).norm() * allg(f).norm())))
    print('  worst cos', [(k, round(cos[k], 3)) for k in ks[:5]], 'median cos', sorted(cos.values())[len(cos) // 2], 'median relL2', sorted(rl.values())[len(rl) // 2])
