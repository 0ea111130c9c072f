#!/usr/bin/env python3
"""MedFormer training steps with the fusion transformer's attention core on csrc/token_attn.hip vs the ATen chain: per-step losses of the
same seeded run (equal up to the reassociation of one soft-max; later steps drift apart the way any two summation orders do under AdamW).
Usage: python tools/token_attn_ab.py [dtype] [steps]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3 import medformer_utils as mu
from rsuper_amd.model.dim3.medformer import MedFormer
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
lf.SANITY_CHECKS = False
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
bt = synth.batch(B, S, classes, ['mask'] * B, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                           ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                           classification_branch=False, ema=True, ema_alpha=0.99)
out = {}
for hip in (True, False):
    mu.HIP_TOKEN_ATTN = hip
    torch.manual_seed(0)
    net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                    num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True, compute_dtype=dtype).to(dev)
    ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    losses, g1 = [], None
    for i in range(steps):
        loss, _ = train_step(net, ema, opt, batch, largs, classes, i)
        losses.append(float(loss['overall']))
        if i == 0:
            g1 = torch.cat([p.grad.flatten().double() for p in net.parameters() if p.grad is not None])
    out[hip] = (losses, g1)
    del net, ema, opt
a, b = out[True], out[False]
print(f'# {dtype}: loss per step, token_attn.hip | ATen chain')
for i, (x, y) in enumerate(zip(a[0], b[0])):
    print(f'step {i}: {x:.7f} | {y:.7f}   diff {abs(x - y):.2e}')
cos = float((a[1] * b[1]).sum() / (a[1].norm() * b[1].norm()))
print(f'first-step gradient: cosine {cos:.9f}, relative L2 difference {float((a[1] - b[1]).norm() / b[1].norm()):.3e}')
