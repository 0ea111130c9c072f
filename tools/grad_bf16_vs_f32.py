#!/usr/bin/env python3
"""How close are the bf16-mode gradients of the REAL loss to the f32-mode ones?  Same initial weights, same batch (config 2: B = 2, 96^3, 26 classes,
segmentation loss), one forward + backward in each arithmetic mode: logits relative L2, cosine / relative L2 of all parameter gradients together and per
tensor.  Usage: grad_bf16_vs_f32.py [unet|medformer]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth
from rsuper_amd.training import losses_foundation as lf
lf.SANITY_CHECKS = False
which = sys.argv[1] if len(sys.argv) > 1 else 'unet'
dev = 'cuda'; B, S = 2, 96; classes = synth.PANTS_CLASSES
bt = synth.batch(B, S, classes, ['mask'] * B, seed=7, diam_range=(5.0, 40.0), max_tumors=3)
batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
             unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
             volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))
largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2, ball_bce_weight=1.0,
                           ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)
out = {}
for mode in ('f32', 'bf16'):
    torch.manual_seed(0)
    if which == 'medformer':
        from rsuper_amd.model.dim3.medformer import MedFormer
        net = MedFormer(1, len(classes), base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
                        num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True, compute_dtype=mode).to(dev)
    else:
        from rsuper_amd.model.dim3.unet import UNet
        net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=mode).to(dev)
    res = net(batch['image'])
    loss = lf.calculate_loss(model_output=res, label=batch['label'], unk_voxels=batch['unk_channels'], args=largs, matcher=None, chosen_segment_mask=batch['mask'],
                             tumor_volumes_report=batch['volumes'], tumor_diameters=batch['diameters'], classes=classes, input_tensor=batch['image'])
    loss['overall'].backward(); torch.cuda.synchronize()
    seg = res['segmentation']; seg = seg[0] if isinstance(seg, (list, tuple)) else seg
    out[mode] = (seg.detach().double(), {k: p.grad.double() for k, p in net.named_parameters()}, float(loss['overall']))
    del net, res, loss
f, b = out['f32'], out['bf16']
l2 = lambda u, v: float((u - v).norm() / v.norm())
allg = lambda t: torch.cat([v.flatten() for v in t[1].values()])
cos = {k: float((b[1][k] * f[1][k]).sum() / (b[1][k].norm() * f[1][k].norm()).clamp_min(1e-300)) for k in f[1]}
ks = sorted(cos, key=cos.get)
print(f'{which}: loss f32 {f[2]:.6f} bf16 {b[2]:.6f}; logits relL2 {l2(b[0], f[0]):.4f}; all gradients: relL2 {l2(allg(b), allg(f)):.4f}, cosine '
      f'{float((allg(b) * allg(f)).sum() / (allg(b).norm() * allg(f).norm())):.4f}; per tensor cosine: min {cos[ks[0]]:.3f} ({ks[0]}), median {sorted(cos.values())[len(cos) // 2]:.3f}')
