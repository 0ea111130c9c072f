#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one training step (is the step launch-bound?)."""
import argparse, os, sys, time
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
for trial in range(3):
    K = 20
    t0 = time.perf_counter()
    for i in range(K):
        train_step(net, ema, opt, batch, largs, classes, 5 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, total {1e3 * (t2 - t0) / K:.2f} ms/step')
# host-only cost: same loop with the GPU idle between steps
ts = []
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step(net, ema, opt, batch, largs, classes, 100 + i)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f'host enqueue with empty queue: {1e3 * sorted(ts)[len(ts) // 2]:.2f} ms/step')

# hipGraph replay (rsuper_amd/graph.py): host cost of one step = a few small copies + one graph launch
from rsuper_amd.graph import GraphedTrainStep
stepper = GraphedTrainStep(net, ema, opt, largs, classes, warmup=3)
for i in range(6):
    stepper(batch, 200 + i)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K):
    stepper(batch, 300 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'hipGraph replay: host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, total {1e3 * (t2 - t0) / K:.2f} ms/step')
ts = []
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stepper(batch, 400 + i)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f'hipGraph replay, host enqueue with empty queue: {1e3 * sorted(ts)[len(ts) // 2]:.3f} ms/step')
