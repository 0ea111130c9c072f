#!/usr/bin/env python3
"""Per-launch HBM rate of the InstanceNorm-backward tail (`in_bwd_finalize`, csrc/unet_misc.hip) inside the headline training step: every call of one step
is bracketed with no-fence HIP events; bytes = (g read + x read + dx written [+ residual gradient read]) x 2 B per element.  VERDICT r05 item 4 asked for
the per-launch GB/s.  Usage: python tools/in_bwd_rates.py"""
import argparse, os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import synth
from rsuper_amd.hip import ops, lib
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
lf.SANITY_CHECKS = False
L = lib.lib()
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
torch.manual_seed(0)
net = UNet(1, 32, num_classes=len(classes), compute_dtype='bf16').to(dev)
ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
bt = synth.batch(B, S, classes, ['mask'] * B, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), **{k: torch.from_numpy(bt[k]).to(dev) for k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')})
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2, ball_bce_weight=1.0,
                           ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)
for i in range(5):
    train_step(net, ema, opt, batch, largs, classes, i)
rec = []
orig = ops.in_bwd_finalize


def ev():
    h = ctypes.c_void_p(); L.rsuper_timer_event_create(ctypes.byref(h)); return h


def timed(g, x, gm, out_C, add1=None):
    a, b = ev(), ev()
    L.rsuper_timer_event_record(a, torch.cuda.current_stream().cuda_stream)
    out = orig(g, x, gm, out_C, add1)
    L.rsuper_timer_event_record(b, torch.cuda.current_stream().cuda_stream)
    rec.append((tuple(out.shape), add1 is not None, a, b))
    return out


ops.in_bwd_finalize = timed
train_step(net, ema, opt, batch, largs, classes, 5)
torch.cuda.synchronize()
tot_us = tot_b = 0.0
print(f'# in_bwd_finalize inside one headline step (B = 2, 96^3, bf16): {len(rec)} launches')
print(f'{"output (N, D, H, W, C)":28s} {"+res":>4s} {"MB":>8s} {"us":>8s} {"TB/s":>6s}')
for shp, res, a, b in rec:
    ms = ctypes.c_float(); L.rsuper_timer_event_elapsed_ms(a, b, ctypes.byref(ms))
    n = 1
    for v in shp:
        n *= v
    by = n * 2 * (4 if res else 3)
    tot_us += ms.value * 1e3; tot_b += by
    print(f'{str(shp):28s} {int(res):4d} {by / 1e6:8.1f} {ms.value * 1e3:8.1f} {by / ms.value / 1e9:6.2f}')
print(f'# total {tot_b / 1e9:.2f} GB in {tot_us / 1e3:.3f} ms = {tot_b / tot_us / 1e6:.2f} TB/s (event brackets include ~1-2 us of launch latency each)')
