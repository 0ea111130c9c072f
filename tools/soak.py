#!/usr/bin/env python3
"""Soak run: N training steps of the bench workload, checks finite losses and a stable allocator footprint."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=300); ap.add_argument('--report', action='store_true')
a = ap.parse_args()
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
torch.manual_seed(0)
net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='bf16').to(dev)
ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
kinds = (['mask', 'report'] * B)[:B] if a.report else ['mask'] * B
bt = synth.batch(B, S, classes, kinds, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_both' if a.report else 'ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0,
                           report_volume_loss_basic=0.1 if a.report else 0.0, volume_loss_tolerance=0.2, ball_bce_weight=1.0, ball_dice_weight=1.0,
                           ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)
t0 = time.time(); mem = []
for i in range(a.steps):
    la, gn = train_step(net, ema, opt, batch, largs, classes, i)
    if i % 50 == 0 or i == a.steps - 1:
        v = float(la['overall'].detach()); g = float(gn)
        mem.append(torch.cuda.max_memory_allocated() / 2 ** 30)
        print(f'step {i:4d} loss {v:.4f} gnorm {g:.3f} max_alloc {mem[-1]:.2f} GiB reserved {torch.cuda.memory_reserved() / 2 ** 30:.2f} GiB', flush=True)
        assert v == v and g == g, 'NaN'
torch.cuda.synchronize()
print(f'done: {a.steps} steps in {time.time() - t0:.1f} s; max_alloc growth after warm-up {mem[-1] - mem[1]:.3f} GiB')
