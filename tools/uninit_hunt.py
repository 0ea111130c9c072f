#!/usr/bin/env python3
"""Every torch.empty filled with NaN (torch.utils.deterministic.fill_uninitialized_memory): a kernel that reads a buffer element nobody wrote
turns the loss / a gradient into NaN.  Runs training steps of MedFormer (shipped config, 128^3, 42 classes, mask + report batch) and of the
UNet (96^3).  Usage: python tools/uninit_hunt.py [medformer|unet] [dtype]"""
import argparse, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
from rsuper_amd.train_ddp import train_step, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
which = sys.argv[1] if len(sys.argv) > 1 else 'medformer'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dev = 'cuda'
if which == 'medformer':
    from rsuper_amd.model.dim3.medformer import MedFormer
    classes = synth.MASK42_CLASSES; S = 128
    net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                    num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True, compute_dtype=dtype).to(dev)
else:
    from rsuper_amd.model.dim3.unet import UNet
    classes = synth.PANTS_CLASSES; S = 96
    net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=dtype, pool=(which != 'nopool')).to(dev)
torch.manual_seed(0)
bt = synth.batch(2, S, classes, ['mask', 'report'], seed=11, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(2, S, seed=99)).to(dev), **{k: torch.from_numpy(v).to(dev) for k, v in bt.items()})
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                           ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                           classification_branch=False, ema=True, ema_alpha=0.99)
ema = make_ema(net); opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
# forward hooks: first module whose output holds a NaN
bad = []
def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for o in outs:
            t = getattr(o, 't', o)
            if torch.is_tensor(t) and t.is_floating_point() and not bad and torch.isnan(t).any():
                bad.append(name)
    return f
for n, m in net.named_modules():
    m.register_forward_hook(hook(n or type(m).__name__))
for step in range(3):
    la, gn = train_step(net, ema, opt, batch, largs, classes, step)
    vals = {k: float(v.detach()) for k, v in la.items()}
    nan_grads = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f'step {step}: overall {vals["overall"]:.6f} grad norm {float(gn):.4f} finite {all(math.isfinite(v) for v in vals.values())}; first NaN module: {bad[:1]}; '
          f'non-finite gradients: {nan_grads[:4]}', flush=True)
