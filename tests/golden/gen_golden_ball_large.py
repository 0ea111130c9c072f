#!/usr/bin/env python3
"""Golden fixtures for the Ball path at the diameters the benchmark's report batch draws: imports the UNMODIFIED
reference (read-only at /root/reference) on CPU, like gen_golden.py, and writes tests/golden/ball_large.npz.

    python tests/golden/gen_golden_ball_large.py

A. isolate_tumor (losses_foundation.py:1387-1532) on synth.ball_case(name): ball kernels of edge 19 ... 51 (d = 15 ... 40), the clipped ball
   + growth loop (:1450-1461), the volume rewrite (:1431-1433), the dilation rounds (:1513-1522).  Masks are stored bit-packed.
B. calculate_loss (:685-1076) at 48^3 / 64^3 with THREE tumours per report sample (d = 31 / 21 / 15 and 40 / 21 / 15), `ball_dice_both`
   and `ball_dice_last` (deep supervision), every returned key + the input gradient.  Inputs come from synth seeds; only outputs are stored.
C. calculate_loss at FULL size (96^3, 26 classes): the exact batch `bench.py --report` builds, and one with d = 40 / 31 / 21.
Every case is also evaluated with oracle/losses_oracle.py here and the comparison printed: a case whose result depends on WHICH exact zeros
torch.topk picks (implementation-defined, differs between torch's CPU and GPU kernels) is marked `<name>_tie_dependent = 1`.
"""
import os
import sys
import contextlib
import io
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth  # noqa: E402
import gen_golden as gg  # noqa: E402

LOSS_CASES = synth.BALL_LOSS_CASES
loss_case_inputs = synth.ball_loss_case_inputs


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    os.chdir(tempfile.mkdtemp())
    _, _, lf, _ = gg.import_reference()
    from oracle import losses_oracle as lo
    t, pack = gg.t, gg.pack
    out = {}
    for name in synth.BALL_CASES:
        x, d, vol = synth.ball_case(name)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m, ms, mb = lf.isolate_tumor(t(x), diameter=d, gaussian=True, gaussian_std=1.5, tumor_volume=vol, diameter_margin=0.2, volume_margin=0.2)
        log = buf.getvalue()
        om = lo.isolate_tumor(t(x), d, vol, 0.2, 0.2, 1.5)
        diff = [int((a != b).sum()) for a, b in zip((m, ms, mb), om[:3])]
        for key, v in (('m', m), ('s', ms), ('b', mb)):
            out[f'iso_{name}_{key}'] = pack(v.numpy())
        out[f'iso_{name}_sums'] = np.array([m.sum().item(), ms.sum().item(), mb.sum().item()], np.float64)
        out[f'iso_{name}_loops'] = np.array([log.count('Increasing ball size'), log.count('dilating tumor mask')], np.int64)
        out[f'iso_{name}_tie_dependent'] = np.array([int(any(diff))], np.int64)
        print(f'iso {name}: edge {x.shape[0]} d {d} vol {vol:.0f} sums {out[f"iso_{name}_sums"]} loops {out[f"iso_{name}_loops"]} oracle diff {diff}')

    for tag in LOSS_CASES:
        classes, bt, lg0, lg1, loss, deep = loss_case_inputs(tag)
        args = gg.make_args(loss=loss)
        a, b = t(lg0).requires_grad_(True), t(lg1).requires_grad_(True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = lf.calculate_loss(model_output={'segmentation': [a, b] if deep else a}, label=t(bt['label']).long(),
                                    unk_voxels=t(bt['unk_channels']).float(), args=args, matcher=None, chosen_segment_mask=t(bt['mask']).float(),
                                    tumor_volumes_report=t(bt['volumes']), tumor_diameters=t(bt['diameters']), classes=classes, input_tensor=None)
        res['overall'].backward()
        log = buf.getvalue()
        for k, v in res.items():
            out[f'{tag}_{k}'] = np.array(v.detach().item() if torch.is_tensor(v) else v, np.float64)
        out[f'{tag}_keys'] = np.array(sorted(res.keys()))
        out[f'{tag}_g0_sub'], _ = synth.subsample(a.grad.numpy(), 8192)
        out[f'{tag}_g0_summary'] = synth.summary(a.grad.numpy())
        if deep:
            out[f'{tag}_g1_sub'], _ = synth.subsample(b.grad.numpy(), 8192)
        out[f'{tag}_loops'] = np.array([log.count('Increasing ball size'), log.count('dilating tumor mask')], np.int64)
        a2, b2 = t(lg0).requires_grad_(True), t(lg1).requires_grad_(True)
        ores = lo.calculate_loss({'segmentation': [a2, b2] if deep else a2}, t(bt['label']), t(bt['unk_channels']), args, t(bt['mask']),
                                 t(bt['volumes']), t(bt['diameters']), classes)
        ores['overall'].backward()
        worst = max(abs(float(ores[k]) - float(res[k])) for k in res)
        gerr = float((a2.grad - a.grad).abs().max() / a.grad.abs().max())
        print(f'loss {tag}: ' + ' '.join(f'{k}={float(v):.6f}' for k, v in res.items()) + f' loops {out[f"{tag}_loops"]} | oracle: max key diff {worst:.2e} grad rel {gerr:.2e}')
    # C. full size: the benchmark's own report batch (and one with d = 40 / 31 / 21) through the reference's calculate_loss
    for tag in synth.FULLSIZE_REPORT_CASES:
        classes, bt, lg = synth.fullsize_report_case(tag)
        args = gg.make_args(loss='ball_dice_both')
        a = t(lg).requires_grad_(True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = lf.calculate_loss(model_output={'segmentation': a}, label=t(bt['label']).long(), unk_voxels=t(bt['unk_channels']).float(), args=args,
                                    matcher=None, chosen_segment_mask=t(bt['mask']).float(), tumor_volumes_report=t(bt['volumes']),
                                    tumor_diameters=t(bt['diameters']), classes=classes, input_tensor=None)
        res['overall'].backward()
        log = buf.getvalue()
        for k, v in res.items():
            out[f'{tag}_{k}'] = np.array(v.detach().item(), np.float64)
        out[f'{tag}_keys'] = np.array(sorted(res.keys()))
        out[f'{tag}_g0_sub'], _ = synth.subsample(a.grad.numpy(), 8192)
        li = classes.index('pancreatic_lesion')
        out[f'{tag}_g0_lesion_sub'], _ = synth.subsample(a.grad.numpy()[1, li], 8192)        # the plane the ball loss writes
        out[f'{tag}_g0_summary'] = synth.summary(a.grad.numpy())
        out[f'{tag}_loops'] = np.array([log.count('Increasing ball size'), log.count('dilating tumor mask')], np.int64)
        print(f'full {tag}: ' + ' '.join(f'{k}={float(v):.6f}' for k, v in res.items()) + f' loops {out[f"{tag}_loops"]}')
    np.savez_compressed(os.path.join(HERE, 'ball_large.npz'), **out)
    print('ball_large.npz', len(out))


if __name__ == '__main__':
    main()
