#!/usr/bin/env python3
"""cProfile of the host side of one training step (where do the ~10 ms of Python per step go?).  Usage: host_profile.py [medformer]"""
import argparse, cProfile, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
lf.SANITY_CHECKS = False
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
if len(sys.argv) > 1 and sys.argv[1] == 'medformer':
    from rsuper_amd.model.dim3.medformer import MedFormer
    net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                    num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True,
                    compute_dtype='bf16').to(dev)
else:
    net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='bf16').to(dev)
ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
bt = synth.batch(B, S, classes, ['mask'] * B, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                           ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                           classification_branch=False, ema=True, ema_alpha=0.99)
for i in range(5):
    train_step(net, ema, opt, batch, largs, classes, i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    train_step(net, ema, opt, batch, largs, classes, 5 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(32)
