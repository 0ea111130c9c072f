#!/usr/bin/env python3
"""Golden fixture for the N > 1 parity definition of SURVEY.md section 8(e): "parity for N GPUs is defined against N independent reference
ranks averaged, not against one big batch".  Imports the UNMODIFIED reference (read-only at /root/reference) on CPU, like gen_golden.py:

    python tests/golden/gen_golden_ddp.py        ->  tests/golden/ddp2.npz

Two ranks of train_ddp.py's step (train_ddp.py:308-349: forward, calculate_loss('ball_dice_both'), backward) on the tiny UNet from identical
weights, rank r on ITS OWN batch (synth.ddp_rank_batch(r): image seed 4321 + r, batch seed 7 + r, one mask + one report sample) -- what DDP's
gradient all-reduce(mean) (train_ddp.py:623-668) must reproduce is the element-wise mean of the two ranks' gradients.  Stored: every loss key per
rank, and per parameter the summary of each rank's gradient and the strided subsample + summary of the mean."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402
import gen_golden as gg  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    os.chdir(tempfile.mkdtemp())
    unet_mod, _, lf, _ = gg.import_reference()
    t = gg.t
    classes = synth.TINY_CLASSES
    out, grads = {}, []
    for r in range(2):
        net = unet_mod.UNet(1, 8, num_classes=len(classes), scale=[2, 2, 2, 2], kernel_size=[3, 3, 3, 3, 3], block='BasicBlock', norm='in')
        gg.load_sd(net, seed=3)
        img, bt = synth.ddp_rank_batch(r)
        res = net(t(img))
        with gg.quiet():
            la = lf.calculate_loss(model_output={'segmentation': res}, label=t(bt['label']).long(), unk_voxels=t(bt['unk_channels']).float(),
                                   args=gg.make_args(loss='ball_dice_both'), matcher=None, chosen_segment_mask=t(bt['mask']).float(),
                                   tumor_volumes_report=t(bt['volumes']), tumor_diameters=t(bt['diameters']), classes=classes, input_tensor=t(img))
        la['overall'].backward()
        for k, v in la.items():
            out[f'r{r}_{k}'] = np.array(v.item(), np.float64)
        grads.append({k: p.grad.numpy().copy() for k, p in net.named_parameters()})
    for k in grads[0]:
        mean = 0.5 * (grads[0][k] + grads[1][k])
        out[f'mean_g_{k}_sub'], _ = synth.subsample(mean, 2048)
        for tag, g in (('r0', grads[0][k]), ('r1', grads[1][k]), ('mean', mean)):
            out[f'{tag}_g_{k}_summary'] = synth.summary(g)
    out['param_names'] = np.array(sorted(grads[0]))
    np.savez_compressed(os.path.join(HERE, 'ddp2.npz'), **out)
    print('ddp2.npz', len(out), {k: float(out[k]) for k in out if k.endswith('overall')})


if __name__ == '__main__':
    main()
