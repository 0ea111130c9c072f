#!/usr/bin/env python3
"""A few MedFormer training steps (config/abdomenatlas_ufo/medformer_3d.yaml at 96^3, B = 2, 26 classes) -- the target of
`rocprofv3 --kernel-trace` runs for SURVEY 8f-1.  Usage: python tools/medformer_step.py [steps] [dtype] [graph|report|netgraph|netgraph-report]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.medformer import MedFormer
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
lf.SANITY_CHECKS = False
if os.environ.get('MF_BLAS'):
    torch.backends.cuda.preferred_blas_library(os.environ['MF_BLAS'])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
torch.manual_seed(0)
net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True, compute_dtype=dtype).to(dev)
ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
report = len(sys.argv) > 3 and sys.argv[3] in ('report', 'netgraph-report')       # report supervision on: one mask + one report sample (eager only)
bt = synth.batch(B, S, classes, ['mask', 'report'] if report else ['mask'] * B, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1 if report else 0.0, volume_loss_tolerance=0.2,
                           ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                           classification_branch=False, ema=True, ema_alpha=0.99)
step_fn = lambda b, i: train_step(net, ema, opt, b, largs, classes, i)
if len(sys.argv) > 3 and sys.argv[3] == 'graph':                  # hipGraph replay: the GPU-side time of the step without the host's launch cost
    from rsuper_amd.graph import GraphedTrainStep
    step_fn = GraphedTrainStep(net, ema, opt, largs, classes, warmup=2)
    dtype += ' graph'
if len(sys.argv) > 3 and sys.argv[3].startswith('netgraph'):   # forward / backward graphs around the eager loss and optimiser
    from rsuper_amd.graph import GraphedNetwork
    gnet = GraphedNetwork(net, warmup=2)
    step_fn = lambda b, i: train_step(gnet, ema, opt, b, largs, classes, i)
    dtype += ' netgraph'
warm = 16 if len(sys.argv) > 3 and (sys.argv[3].startswith('netgraph') or sys.argv[3] == 'graph') else 4      # GraphedNetwork verifies itself at replays 1 and 12: keep them out of the timing
for i in range(warm):
    step_fn(batch, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss, _ = step_fn(batch, warm + i)
torch.cuda.synchronize()
print(f'medformer {dtype}{" report" if report else ""}: {1e3 * (time.perf_counter() - t0) / steps:.1f} ms/step, loss {float(loss["overall"]):.6f}, '
      f'peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
