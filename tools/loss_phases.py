#!/usr/bin/env python3
"""Where does the report supervision spend its time?  Times the phases of calculate_loss on fixed logits (config 3: one mask sample + one
report sample, 96^3, 26 classes), each bracketed by a device synchronisation: ball search (host-synchronous), everything else of the forward,
and the backward down to d(logits).  Usage: python tools/loss_phases.py [reps]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.training import losses_foundation as lf
lf.SANITY_CHECKS = False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
bt = synth.batch(B, S, classes, ['mask', 'report'], seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = {k: torch.from_numpy(bt[k]).to(dev) for k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
torch.manual_seed(0)
logits0 = torch.randn(B, len(classes), S, S, S, device=dev) * 2.0
largs = argparse.Namespace(loss='ball_dice_both', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                           ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                           classification_branch=False, ema=True, ema_alpha=0.99)
sargs = argparse.Namespace(**{**vars(largs), 'loss': 'ball_dice_last', 'report_volume_loss_basic': 0.0})
plans_t = [0.0]
orig = lf._ball_plans
def timed_plans(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig(*a, **k)
    torch.cuda.synchronize(); plans_t[0] += time.perf_counter() - t
    return r
lf._ball_plans = timed_plans


def run(args):
    tf = tb = 0.0
    plans_t[0] = 0.0
    for i in range(reps + 2):
        if i == 2:
            tf = tb = 0.0; plans_t[0] = 0.0
        x = logits0.clone().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = lf.calculate_loss(model_output={'segmentation': x}, label=batch['label'], unk_voxels=batch['unk_channels'], args=args, matcher=None,
                                 chosen_segment_mask=batch['mask'], tumor_volumes_report=batch['volumes'], tumor_diameters=batch['diameters'],
                                 classes=classes)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss['overall'].backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1
    return 1e3 * tf / reps, 1e3 * plans_t[0] / reps, 1e3 * tb / reps


for name, a in (('segmentation only', sargs), ('report supervision', largs)):
    f, p, b = run(a)
    print(f'{name:20s} forward {f:6.2f} ms (ball search {p:5.2f}, rest {f - p:5.2f})   backward {b:5.2f} ms')
