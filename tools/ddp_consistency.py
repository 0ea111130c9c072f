#!/usr/bin/env python3
"""GradReducer (wrap_ddp on a 1-rank RCCL group) vs the plain module: the same full-size training run with report supervision, 30 steps, losses and
gradient norms equal at every step (mean over one rank = identity; exercises the flat buckets, the hooks and the collectives).
Usage: python tools/ddp_consistency.py [medformer]"""
import argparse, os, sys
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.model.dim3.medformer import MedFormer
from rsuper_amd.train_ddp import make_ema, train_step, wrap_ddp
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
from rsuper_amd.hip import ops
lf.SANITY_CHECKS = False
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
bt = synth.batch(B, S, classes, ['mask', 'report'], seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                           ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                           classification_branch=False, ema=True, ema_alpha=0.99)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29547')
dist.init_process_group(backend='nccl', rank=0, world_size=1)
hist = []
for wrapped in (False, True):
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'medformer':
        net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                        num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True, compute_dtype='bf16').to(dev)
    else:
        net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='bf16').to(dev)
    ema = make_ema(net)
    model = wrap_ddp(net, 0) if wrapped else net
    opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    ls = []
    for i in range(30):
        loss, gn = train_step(model, ema, opt, batch, largs, classes, i)
        ls.append((float(loss['overall'].detach()), float(gn)))
    hist.append(ls)
    red = getattr(net, '_rsuper_reducer', None)
    if red is not None:
        red.remove()
    ops.GRAD_DEST = None
dist.destroy_process_group()
bad = [(i, a, b) for i, (a, b) in enumerate(zip(*hist)) if a != b]
print('GradReducer (1-rank RCCL) vs plain, full-size network with report supervision, 30 steps:', 'identical' if not bad else f'first difference {bad[0]}')
