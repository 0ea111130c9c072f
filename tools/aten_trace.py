#!/usr/bin/env python3
"""Which lines of this package still launch ATen kernels inside one training step?  Runs a few config-2 steps under
torch.profiler (with_stack) and prints, per (aten op, innermost rsuper_amd source line), the launches per step and the device
time -- the work list for VERDICT item 4 ("kill the ATen leftovers").  Usage: python tools/aten_trace.py [--report] [--medformer] [--steps 3]"""
import argparse, collections, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf

ap = argparse.ArgumentParser()
ap.add_argument('--report', action='store_true')
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--medformer', action='store_true', help='the shipped MedFormer configuration instead of the UNet')
a = ap.parse_args()
lf.SANITY_CHECKS = False
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
if a.medformer:
    from rsuper_amd.model.dim3.medformer import MedFormer
    net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                    num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True,
                    compute_dtype='bf16').to(dev)
else:
    net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='bf16').to(dev)
ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
kinds = (['mask', 'report'] * B)[:B] if a.report else ['mask'] * B
bt = synth.batch(B, S, classes, kinds, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1 if a.report else 0.0,
                           volume_loss_tolerance=0.2, ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False,
                           stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)
for i in range(4):
    train_step(net, ema, opt, batch, largs, classes, i)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = ('aten.empty', 'aten.view', 'aten._unsafe_view', 'aten.reshape', 'aten.as_strided', 'aten.detach', 'aten.alias', 'aten.select',
        'aten.slice', 'aten.t.', 'aten.transpose', 'aten.permute', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten._local_scalar_dense',
        'aten.lift_fresh', 'aten.empty_like', 'aten.empty_strided', 'aten.new_empty', 'aten.unbind', 'aten.split', 'aten.is_same_size')
calls = collections.Counter()


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            tensors = [x for x in args if isinstance(x, torch.Tensor)]
            if any(t.is_cuda for t in tensors) or not tensors:
                where = 'autograd engine'
                for fr in reversed(traceback.extract_stack()):
                    if 'r-super_amd' in fr.filename or fr.filename.endswith('bench.py'):
                        where = '%s:%d %s' % (fr.filename.replace(ROOT + '/', ''), fr.lineno, fr.name)
                        break
                shape = tuple(tensors[0].shape) if tensors else ()
                calls[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


with Trace():
    for i in range(a.steps):
        train_step(net, ema, opt, batch, largs, classes, 100 + i)
torch.cuda.synchronize()
print('# dispatcher view: ops per step on device tensors (views/allocations skipped), with the innermost package frame')
for (name, where, shape), n in sorted(calls.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f'{n / a.steps:6.1f}  {name:34s} {str(shape):28s} {where}')

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(a.steps):
        train_step(net, ema, opt, batch, largs, classes, 4 + i)
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or not ev.kernels:
        continue
    # leaf ops only: the op that owns the kernels directly
    where = '?'
    for fr in (ev.stack or []):
        if 'r-super_amd' in fr or 'rsuper_amd' in fr or 'bench.py' in fr:
            where = fr.replace(ROOT + '/', '')
            break
    k = (ev.name, where)
    agg[k][0] += len(ev.kernels)
    agg[k][1] += sum(kk.duration for kk in ev.kernels)
tot_l = sum(v[0] for v in agg.values()) / a.steps
tot_t = sum(v[1] for v in agg.values()) / a.steps
print(f'# ATen launches / step: {tot_l:.1f}, device time {tot_t:.1f} us / step ({"config 3" if a.report else "config 2"})')
print(f'{"op":28s} {"launch/step":>11s} {"us/step":>9s}  where')
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{name:28s} {n / a.steps:11.1f} {t / a.steps:9.1f}  {where}')

# in-place / internal ops (copy_, fill_, add_ ...) never reach the dispatcher hook above with a Python frame: name them by the chain of
# enclosing profiler ranges (autograd node / parent aten op) and their input shapes
chains = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in ('aten::copy_', 'aten::fill_', 'aten::add_', 'aten::zero_') or ev.device_time_total <= 0 or not ev.kernels:
        continue
    names, q = [], ev.cpu_parent
    while q is not None and len(names) < 4:
        names.append(q.name)
        q = q.cpu_parent
    k = (ev.name, ' < '.join(names), str(ev.input_shapes[:2]))
    chains[k][0] += len(ev.kernels)
    chains[k][1] += sum(kk.duration for kk in ev.kernels)
print('# in-place leaves by enclosing range')
for (name, chain, shp), (n, t) in sorted(chains.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{name:14s} {n / a.steps:7.1f} {t / a.steps:9.1f} us  {shp:40s} {chain}')

# library GEMMs by operand shapes (which 1x1x1 convolutions / weight gradients are slow?)
gemm = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ('aten::mm', 'aten::addmm', 'aten::bmm') and ev.device_time_total > 0 and ev.kernels:
        k = (ev.name, str(ev.input_shapes[:3]))
        gemm[k][0] += 1
        gemm[k][1] += sum(kk.duration for kk in ev.kernels)
print('# GEMMs by shape')
for (name, shp), (n, t) in sorted(gemm.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{name:12s} {n / a.steps:6.1f} calls {t / a.steps:9.1f} us/step {t / n:8.1f} us each  {shp}')
