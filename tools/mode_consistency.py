#!/usr/bin/env python3
"""Eager vs replayed execution of the SAME training run, step by step: losses and gradient norm must be equal at every step (the step is
bit-reproducible), and at the first difference the gradient tensors that differ are listed.  This comparison found the two hipGraph defects of
round 2.  Usage: mode_consistency.py [unet|medformer] [size] [classes] [steps] [seg|report] [step|net]
  step: whole-step graph (GraphedTrainStep, segmentation-only supervision);  net: forward / backward graphs (GraphedNetwork)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.train_ddp import make_ema, train_step
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
from rsuper_amd.graph import GraphedTrainStep, GraphedNetwork
lf.SANITY_CHECKS = False
which = sys.argv[1] if len(sys.argv) > 1 else 'medformer'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
ncls = int(sys.argv[3]) if len(sys.argv) > 3 else 26
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
report = (sys.argv[5] if len(sys.argv) > 5 else 'seg') == 'report'
form = sys.argv[6] if len(sys.argv) > 6 else 'net'
dev = 'cuda'; B = 2
classes = synth.MASK42_CLASSES if ncls == 42 else synth.PANTS_CLASSES
bt = synth.batch(B, S, classes, ['mask', 'report'] if report else ['mask'] * B, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1 if report else 0.0,
                           volume_loss_tolerance=0.2, ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False,
                           stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)


def build():
    torch.manual_seed(0)
    if which == 'medformer':
        from rsuper_amd.model.dim3.medformer import MedFormer
        return MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                         num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True,
                         compute_dtype='bf16').to(dev)
    from rsuper_amd.model.dim3.unet import UNet
    return UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='bf16').to(dev)


hist, grads = [], []
for graphed in (False, True):
    net = build()
    ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    st = GraphedTrainStep(net, ema, opt, largs, classes, warmup=2) if (graphed and form == 'step') else None
    f = GraphedNetwork(net, warmup=2) if (graphed and form == 'net') else net
    ls, gs = [], []
    for i in range(steps):
        loss, gn = st(batch, i) if st is not None else train_step(f, ema, opt, batch, largs, classes, i)
        ls.append({k: float(v.detach()) for k, v in loss.items()} | {'gn': float(gn)})
        if os.environ.get('MC_DEBUG') == '1' and graphed and 48 <= i <= 55:
            tot = float(opt._total_sq) if getattr(opt, '_total_sq', None) is not None else float('nan')
            ref = float(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None))
            print(f'[mc] step {i}: gn {float(gn)!r} total_sq {tot!r} sqrt {tot ** 0.5 if tot >= 0 else float("nan")!r}; torch sum g^2 {ref!r} sqrt {ref ** 0.5!r}', flush=True)
        gs.append({k: p.grad.detach().clone() for k, p in net.named_parameters()} if i < 24 else None)
    hist.append(ls); grads.append(gs)
    del net, ema, opt, st, f
    torch.cuda.empty_cache()
for i in range(steps):
    a, b = hist[0][i], hist[1][i]
    if a != b:
        print('first difference at step', i, {k: (a[k], b[k]) for k in a if a[k] != b[k]})
        for j in range(i + 1, min(i + 4, steps)):
            print('   step', j, {k: (hist[0][j][k], hist[1][j][k]) for k in hist[0][j] if hist[0][j][k] != hist[1][j][k]})
        if grads[0][i] is not None:
            ga, gb = grads[0][i], grads[1][i]
            bad = [(k, float((ga[k] - gb[k]).abs().max()), float(ga[k].abs().max())) for k in ga if not torch.equal(ga[k], gb[k])]
            print(len(bad), 'of', len(ga), 'gradient tensors differ')
            for k, dd, m in sorted(bad, key=lambda t: -t[1] / max(t[2], 1e-30))[:8]:
                print(f'  {k:60s} max diff {dd:.3e} of max {m:.3e}')
        sys.exit(1)
print(f'{which} {S}^3 {len(classes)} classes {"report" if report else "seg"} supervision, form {form}: eager and replayed runs identical over {steps} steps '
      f'(final loss {hist[0][-1]["overall"]:.6f})')
