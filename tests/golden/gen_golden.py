#!/usr/bin/env python3
"""Generate golden fixtures by importing the UNMODIFIED reference (read-only at
/root/reference) on CPU.  Run in the authoring container only:

    python tests/golden/gen_golden.py

The reference's Python never travels to the GPU box; only the .npz files written
here (inputs are regenerated from tests/golden/synth.py seeds, outputs are stored)
do.  Import shims follow SURVEY.md section 8(c): stub `nibabel` (debug dumps
only) and pre-register `model`/`model.dim3` so MONAI/timm-dependent nets in
model/dim3/__init__.py are bypassed.
"""
import os
import sys
import types
import tempfile
import importlib
import argparse
import contextlib
import io

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = '/root/reference/rsuper_train'


def import_reference():
    nib = types.ModuleType('nibabel')
    nib.Nifti1Image = lambda *a, **k: None
    nib.save = lambda *a, **k: None
    sys.modules['nibabel'] = nib
    sys.path.insert(0, REF)
    for name, sub in (('model', 'model'), ('model.dim3', 'model/dim3')):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    unet = importlib.import_module('model.dim3.unet')
    conv_layers = importlib.import_module('model.dim3.conv_layers')
    lf = importlib.import_module('training.losses_foundation')
    tu = importlib.import_module('training.utils')
    lf.counter = lf.counter2 = lf.counter3 = 99  # silence NIfTI debug dumps
    return unet, conv_layers, lf, tu


def load_sd(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd


def pack(a):
    a = np.asarray(a)
    return np.packbits(a.astype(np.uint8).reshape(-1))


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_args(**kw):
    d = dict(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1,
             volume_loss_tolerance=0.2, ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2,
             multi_ch_tumor=False, stardard_ce_ball=False, classification_branch=False)
    d.update(kw)
    return argparse.Namespace(**d)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    work = tempfile.mkdtemp()
    os.chdir(work)
    unet_mod, cl, lf, tu = import_reference()
    out = {}

    # ------------------------------------------------------------------ A. primitives
    prim = {}
    for d in [1, 3, 5, 7, 8, 10, 15, 31, 40]:
        k = lf.create_ball_kernel(d)
        prim[f'ball_{d}_edge_nnz'] = np.array([k.shape[0], int((k > 0).sum())], np.int64)
    for d in [3, 5, 9]:
        prim[f'gball_{d}'] = lf.create_ball_kernel(d, gaussian=True, gaussian_std=1.5).numpy()
    prim['ball_6p0'] = lf.create_ball_kernel(5 * 1.2).numpy().astype(np.uint8)
    g = synth.rng(11)
    vol = (g.random((2, 3, 20, 20, 20)) < 0.002).astype(np.float32)
    prim['dil_in'] = pack(vol)
    for ks in [1, 2, 3, 5, 7, 9, 13, 31]:
        prim[f'dil_{ks}'] = pack(lf.dilate_volume(t(vol), ks).numpy())
    single = np.zeros((40, 40, 40), np.float32)
    single[20, 20, 20] = 1
    dsingle = lf.dilate_volume(t(single), 31).numpy()
    prim['dil_single31_sum'] = np.array([dsingle.sum()], np.float64)
    unk = (g.random((2, 3, 16, 16, 16)) < 0.01).astype(np.float32)
    unk[:, 0] = 0
    with quiet():
        kv = lf.get_known_voxels(t(unk), t(unk), dilation=5, sanity=False)
    prim['known_in'] = pack(unk)
    prim['known_out'] = pack(kv.numpy())
    xs = np.array([[0., 50., 268., 1000., 5000., 120., 90.]], np.float32)
    ys = np.array([[268.08, 268.08, 268.08, 268.08, 0., 100., 80.]], np.float32)
    for tol in (0.1, 0.2):
        prim[f'dvl_tol{tol}'] = lf.dice_based_volume_loss(t(xs), t(ys), tolerance=tol, E=500).numpy()
    prim['dvl_x'], prim['dvl_y'] = xs, ys
    # DiceLossMultiClass with attached alpha
    p = g.standard_normal((2, 3, 10, 10, 10)).astype(np.float32) * 2
    tg = (g.random((2, 3, 10, 10, 10)) < 0.3).astype(np.float32)
    kn = (g.random((2, 3, 10, 10, 10)) < 0.9).astype(np.float32)
    pt = t(p).requires_grad_(True)
    dl = lf.DiceLossMultiClass(pt, t(tg), t(kn), sigmoid=True)
    dl.backward()
    prim['dice_p'], prim['dice_t'], prim['dice_k'] = p, tg, kn
    prim['dice_loss'] = dl.detach().numpy()
    prim['dice_grad'] = pt.grad.numpy()
    cw = g.uniform(0.5, 2.0, (2, 3)).astype(np.float32)
    pt2 = t(p).requires_grad_(True)
    dl2 = lf.DiceLossMultiClass(pt2, t(tg), t(kn), sigmoid=True, class_weights=t(cw)[:, :, None, None, None])
    dl2.backward()
    prim['dice_cw'] = cw
    prim['dice_loss_cw'] = dl2.detach().numpy()
    prim['dice_grad_cw'] = pt2.grad.numpy()
    # GWRP weights (return_weights + hard_cutoff path)
    S = 12
    pm = np.zeros((S, S, S), np.float32)
    pm[3:8, 4:9, 2:7] = (g.random((5, 5, 5)) < 0.7)
    xg = 1.0 / (1.0 + np.exp(-g.standard_normal((S, S, S)).astype(np.float32)))
    w = lf.GlobalWeightedRankPooling(t(xg * pm + pm), N=t(pm).sum(), c=0.5, return_weights=True, hard_cutoff=True)
    prim['gwrp_x'], prim['gwrp_pm'], prim['gwrp_w'] = xg.astype(np.float32), pm, w.numpy()
    # isolate_tumor
    S = 28
    lg = synth.logits(1, 1, S, seed=5)[0, 0]
    seg = np.zeros((S, S, S), np.float32)
    seg[4:24, 4:24, 4:24] = 1
    xi = (1.0 / (1.0 + np.exp(-lg))).astype(np.float32) * seg
    for name, (dia, volv) in {'a': (7.0, 150.0), 'b': (4.6, 40.0), 'c': (9.0, 300.0)}.items():
        with quiet():
            m, ms, mb = lf.isolate_tumor(t(xi), diameter=dia, gaussian=True, gaussian_std=1.5, tumor_volume=volv,
                                         diameter_margin=0.2, volume_margin=0.2)
        prim[f'iso_{name}_m'], prim[f'iso_{name}_s'], prim[f'iso_{name}_b'] = pack(m.numpy()), pack(ms.numpy()), pack(mb.numpy())
        prim[f'iso_{name}_sums'] = np.array([m.sum().item(), ms.sum().item(), mb.sum().item()], np.float64)
    prim['iso_x'] = xi
    # isolate_tumor near the border (forces ball growth loop, :1450-1461)
    xb = np.zeros((S, S, S), np.float32)
    xb[0:3, 0:3, 0:3] = 0.9
    xb += 0.01 * (1.0 / (1.0 + np.exp(-lg))).astype(np.float32)
    with quiet():
        m, ms, mb = lf.isolate_tumor(t(xb), diameter=9.0, gaussian=True, gaussian_std=1.5, tumor_volume=380.0,
                                     diameter_margin=0.2, volume_margin=0.2)
    prim['iso_border_x'] = xb
    prim['iso_border_m'], prim['iso_border_s'], prim['iso_border_b'] = pack(m.numpy()), pack(ms.numpy()), pack(mb.numpy())
    np.savez_compressed(os.path.join(HERE, 'primitives.npz'), **prim)
    print('primitives.npz', len(prim))

    # ------------------------------------------------------------------ B. blocks
    blk = {}
    import torch.nn as nn
    # *_s2: stride-(2,2,2) first block of down_block(pool=False) (unet_utils.py:38-39); b16_16_s2 on an odd size (9 -> 5)
    for tag, (ci, co, S) in {'b8_16': (8, 16, 12), 'b16_16': (16, 16, 10), 'b24_8': (24, 8, 12), 'b8_16_s2': (8, 16, 12), 'b16_16_s2': (16, 16, 9)}.items():
        st = 2 if tag.endswith('_s2') else 1
        m = cl.BasicBlock(ci, co, kernel_size=[3, 3, 3], norm=nn.InstanceNorm3d, stride=st) if st == 2 else \
            cl.BasicBlock(ci, co, kernel_size=[3, 3, 3], norm=nn.InstanceNorm3d)
        load_sd(m, seed={'b8_16': 1, 'b16_16': 2, 'b24_8': 3, 'b8_16_s2': 4, 'b16_16_s2': 5}[tag])
        So = (S + 1) // 2 if st == 2 else S
        x = synth.rng(40 + ci).standard_normal((2, ci, S, S, S)).astype(np.float32)
        go = synth.rng(50 + co).standard_normal((2, co, So, So, So)).astype(np.float32)
        xt = t(x).requires_grad_(True)
        y = m(xt)
        y.backward(t(go))
        blk[f'{tag}_y'] = y.detach().numpy()
        blk[f'{tag}_dx'] = xt.grad.numpy()
        for k, v in m.named_parameters():
            blk[f'{tag}_dw_{k}'] = v.grad.numpy()
    # pool / upsample primitives used by down_block/up_block
    x = synth.rng(61).standard_normal((1, 8, 8, 8, 8)).astype(np.float32)
    xt = t(x).requires_grad_(True)
    y = torch.nn.functional.max_pool3d(xt, 2)
    go = synth.rng(62).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(t(go))
    blk['pool_y'], blk['pool_dx'] = y.detach().numpy(), xt.grad.numpy()
    x = synth.rng(63).standard_normal((1, 8, 3, 3, 3)).astype(np.float32)
    xt = t(x).requires_grad_(True)
    y = torch.nn.functional.interpolate(xt, size=(6, 6, 6), mode='trilinear', align_corners=True)
    go = synth.rng(64).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(t(go))
    blk['up_y'], blk['up_dx'] = y.detach().numpy(), xt.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'blocks.npz'), **blk)
    print('blocks.npz', len(blk))

    # ------------------------------------------------------------------ C. tiny UNet fwd/bwd (BASELINE config 1 shape)
    un = {}
    classes = synth.TINY_CLASSES
    net = unet_mod.UNet(1, 8, num_classes=len(classes), scale=[2, 2, 2, 2], kernel_size=[3, 3, 3, 3, 3],
                        block='BasicBlock', norm='in')
    sd = load_sd(net, seed=3)
    un['param_checksum'] = np.array([sum(float(np.abs(v).sum()) for v in sd.values())], np.float64)
    S = 48
    img = synth.image(1, S, seed=1234)
    y = net(t(img))
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    y.backward(t(go))
    un['logits_sub'], step = synth.subsample(y.detach().numpy(), 8192)
    un['logits_step'] = np.array([step])
    un['logits_summary'] = synth.summary(y.detach().numpy())
    for k, v in net.named_parameters():
        gnp = v.grad.numpy()
        un[f'g_{k}_head'] = gnp.reshape(-1)[:64].copy()
        un[f'g_{k}_sub'], _ = synth.subsample(gnp, 4096)            # strided over the whole tensor (every tap / channel region)
        un[f'g_{k}_summary'] = synth.summary(gnp)
    # the same network with strided down-sampling instead of MaxPool (UNet(..., pool=False), unet.py:36-39)
    net2 = unet_mod.UNet(1, 8, num_classes=len(classes), scale=[2, 2, 2, 2], kernel_size=[3, 3, 3, 3, 3], block='BasicBlock', norm='in', pool=False)
    load_sd(net2, seed=3)
    y2 = net2(t(img))
    y2.backward(t(go))
    un['nopool_logits_sub'], _ = synth.subsample(y2.detach().numpy(), 8192)
    un['nopool_logits_summary'] = synth.summary(y2.detach().numpy())
    for k, v in net2.named_parameters():
        un[f'nopool_g_{k}_summary'] = synth.summary(v.grad.numpy())
        un[f'nopool_g_{k}_sub'], _ = synth.subsample(v.grad.numpy(), 4096)
    np.savez_compressed(os.path.join(HERE, 'unet_tiny.npz'), **un)
    print('unet_tiny.npz', len(un))

    # ------------------------------------------------------------------ D. calculate_loss
    cl_out = {}
    S, B = 32, 2
    C = len(classes)
    bt = synth.batch(B, S, classes, ['mask', 'report'], seed=7, diam_range=(5.0, 9.0), max_tumors=2)
    lg0 = synth.logits(B, C, S, seed=99)
    lg1 = synth.logits(B, C, S, seed=100)
    cwts = synth.rng(5).uniform(0.5, 2.0, (B, C)).astype(np.float32)

    def run(tag, args, deep, with_report=True, class_weights=None, batch=bt, classes=classes, lg0=lg0, lg1=lg1):
        a = t(lg0).requires_grad_(True)
        b = t(lg1).requires_grad_(True)
        mo = {'segmentation': [a, b] if deep else a}
        if with_report:
            mask_, vols_, dias_ = t(batch['mask']).float(), t(batch['volumes']), t(batch['diameters'])
        else:
            mask_, vols_, dias_ = t(batch['mask']).float(), t(batch['volumes']), t(batch['diameters'])
        with quiet():
            res = lf.calculate_loss(model_output=mo, label=t(batch['label']).long(), unk_voxels=t(batch['unk_channels']).float(),
                                    args=args, matcher=None, chosen_segment_mask=mask_, tumor_volumes_report=vols_,
                                    tumor_diameters=dias_, classes=classes, input_tensor=None,
                                    class_weights=None if class_weights is None else t(class_weights))
        res['overall'].backward()
        for k, v in res.items():
            cl_out[f'{tag}_{k}'] = np.array(v.detach().item() if torch.is_tensor(v) else v, np.float64)
        cl_out[f'{tag}_g0_sub'], st = synth.subsample(a.grad.numpy(), 8192)
        cl_out[f'{tag}_g0_summary'] = synth.summary(a.grad.numpy())
        if deep:
            cl_out[f'{tag}_g1_sub'], _ = synth.subsample(b.grad.numpy(), 8192)
            cl_out[f'{tag}_g1_summary'] = synth.summary(b.grad.numpy())
        cl_out[f'{tag}_keys'] = np.array(sorted(res.keys()))

    run('single_last', make_args(loss='ball_dice_last'), deep=False)
    run('single_both', make_args(loss='ball_dice_both'), deep=False)
    run('single_dice', make_args(loss='dice'), deep=False)
    run('single_ball', make_args(loss='ball'), deep=False)
    run('single_norep', make_args(report_volume_loss_basic=0.0), deep=False)
    run('deep_last', make_args(loss='ball_dice_last'), deep=True)
    run('deep_dice', make_args(loss='dice'), deep=True)
    run('single_both_cw', make_args(loss='ball_dice_both'), deep=False, class_weights=cwts)
    bt2 = synth.batch(B, S, classes, ['healthy', 'mask'], seed=8)
    run('single_both_norpt', make_args(loss='ball_dice_both'), deep=False, batch=bt2)
    # lesion group spanning two channels: get_lesion_channels max-merge (training/losses_foundation.py:218-219)
    mcls = synth.MULTI_CH_CLASSES
    btm = synth.multi_ch_batch(B, S, ['mask', 'report'], seed=7, diam_range=(5.0, 9.0), max_tumors=2)
    run('multi_ch_both', make_args(loss='ball_dice_both'), deep=False, batch=btm, classes=mcls,
        lg0=synth.logits(B, len(mcls), S, seed=199), lg1=synth.logits(B, len(mcls), S, seed=200))
    run('multi_ch_deep_last', make_args(loss='ball_dice_last'), deep=True, batch=btm, classes=mcls,
        lg0=synth.logits(B, len(mcls), S, seed=199), lg1=synth.logits(B, len(mcls), S, seed=200))
    cl_out['cw'] = cwts
    np.savez_compressed(os.path.join(HERE, 'calc_loss.npz'), **cl_out)
    print('calc_loss.npz', len(cl_out))

    # ------------------------------------------------------------------ E. two training steps (train_ddp.py:308-357)
    ts = {}
    import copy
    net = unet_mod.UNet(1, 8, num_classes=C, scale=[2, 2, 2, 2], kernel_size=[3, 3, 3, 3, 3], block='BasicBlock', norm='in')
    load_sd(net, seed=3)
    ema = copy.deepcopy(net)
    for p_ in ema.parameters():
        p_.requires_grad_(False)
    targs = argparse.Namespace(optimizer='adamw', base_lr=6e-4, betas=[0.9, 0.999], weight_decay=0.05)
    opt = tu.get_optimizer(targs, net)
    args = make_args(loss='ball_dice_both')
    S = 32
    img = synth.image(B, S, seed=4321)
    for step_i in range(2):
        opt.zero_grad()
        r = net(t(img))
        with quiet():
            la = lf.calculate_loss(model_output={'segmentation': r}, label=t(bt['label']).long(),
                                   unk_voxels=t(bt['unk_channels']).float(), args=args, matcher=None,
                                   chosen_segment_mask=t(bt['mask']).float(), tumor_volumes_report=t(bt['volumes']),
                                   tumor_diameters=t(bt['diameters']), classes=classes, input_tensor=t(img))
        la['overall'].backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            tu.update_ema_variables(net, ema, 0.99, step_i)
        for k, v in la.items():
            ts[f's{step_i}_{k}'] = np.array(v.item(), np.float64)
        ts[f's{step_i}_gradnorm'] = np.array(gn.item(), np.float64)
        for k in ['inc.conv1.weight', 'down2.conv.1.conv1.conv.weight', 'up4.conv.0.shortcut.conv.weight', 'outc.weight', 'outc.bias']:
            ts[f's{step_i}_p_{k}_head'] = dict(net.named_parameters())[k].detach().numpy().reshape(-1)[:64].copy()
            ts[f's{step_i}_ema_{k}_head'] = dict(ema.named_parameters())[k].detach().numpy().reshape(-1)[:64].copy()
    ts['lr_sched'] = np.array([_lr(tu, e) for e in [0, 1, 3, 5, 6, 50, 99]], np.float64)
    np.savez_compressed(os.path.join(HERE, 'train_step.npz'), **ts)
    print('train_step.npz', len(ts))


def _lr(tu, epoch):
    class O:
        param_groups = [{'lr': 6e-4}]
    o = O()
    return tu.exp_lr_scheduler_with_warmup(o, epoch, 5, 100)


if __name__ == '__main__':
    main()
