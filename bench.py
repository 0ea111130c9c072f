#!/usr/bin/env python3
"""bench.py -- throughput of the R-Super training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one iteration of train_epoch (rsuper_train/train_ddp.py:308-357) on a synthetic batch already resident
in HBM: zero_grad -> UNet forward -> calculate_loss -> backward -> clip_grad_norm_(1.0) -> AdamW -> EMA.
Workload = BASELINE.json configs[1]: full R-Super 3D UNet (base 32, 26 PanTS classes, 40.56 M parameters), bf16,
96^3 patches, batch 2 per GPU, segmentation loss (masked BCE + adaptive-Tversky Dice), report losses off.
Rank 0 prints ONE JSON line; `value` = voxels processed by all ranks / max-over-ranks time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md
# Fabric-side bytes (FETCH_SIZE x2 + WRITE_SIZE) of all conv MFMA launches of ONE default step, from the rocprofv3 PMC
# passes in profiles/r01_pmc_step.md (bf16, base 32, B=2, 96^3, report losses off); other workloads report null.
CONV_TRAFFIC_BYTES_PER_STEP = 20.7e9


def conv_stack_flops(base, S, B):
    """Algorithmic FLOPs of the 43 3x3x3 convolutions of UNet(base) per forward (2*MACs), SURVEY.md section 8(d)."""
    b = base
    ch = [b, 2 * b, 4 * b, 8 * b, 10 * b]
    total = 0.0

    def blk(ci, co, s):
        n = 2.0 * B * s ** 3 * 27
        t = n * ci * co + n * co * co
        if ci != co:
            t += n * ci * co
        return t
    total += blk(b, b, S)
    s = S
    for i in range(4):
        s //= 2
        total += blk(ch[i], ch[i + 1], s) + blk(ch[i + 1], ch[i + 1], s)
    for i in range(4):
        s *= 2
        ci, co = ch[4 - i], ch[3 - i]
        total += blk(ci + co, co, s) + blk(co, co, s)
    return total


def cpu_baseline(args, classes):
    """Oracle (CPU restatement of the reference, kind='port') timed on the host cores on a bounded sample."""
    from oracle import unet_oracle as uo, losses_oracle as lo
    import synth
    ncores = min(os.cpu_count(), 32)      # ATen CPU conv3d stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(ncores)
    B, S = 1, args.size                   # bounded sample (one B=1 step): ~10-30 s of CPU work
    shapes = uo.unet_param_shapes(1, args.base, len(classes))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.fill_state_dict(shapes, 3).items()}
    img = torch.from_numpy(synth.image(B, S, seed=1234))
    bt = synth.batch(B, S, classes, ['mask'] * B, seed=7)
    la = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0,
                            volume_loss_tolerance=0.2, ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2,
                            multi_ch_tumor=False, stardard_ce_ball=False, classification_branch=False)
    t0 = time.time()
    r = uo.unet_forward(sd, img)
    res = lo.calculate_loss({'segmentation': r}, torch.from_numpy(bt['label']), torch.from_numpy(bt['unk_channels']), la,
                            torch.from_numpy(bt['mask']), torch.from_numpy(bt['volumes']), torch.from_numpy(bt['diameters']), classes)
    res['overall'].backward()
    dt = time.time() - t0
    return {'value': B * S ** 3 / dt, 'unit': 'voxels/s', 'cores': ncores, 'kind': 'port',
            'sample': f'1 un-warmed step (fwd + seg loss + bwd, no optimiser) of the fp32 torch-CPU oracle, B={B}, {S}^3, base {args.base}, '
                      f'{len(classes)} classes, {ncores} threads, {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=96)
    ap.add_argument('--batch', type=int, default=2, help='per-GPU batch (bs=2/GPU in BASELINE.json)')
    ap.add_argument('--base', type=int, default=32)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--report', action='store_true', help='config 3: report supervision on (ball_dice_both, 50/50 mask/report batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-ddp', action='store_true', help='wrap in DistributedDataParallel even with one rank (exercises the RCCL reducer path)')
    args = ap.parse_args()

    import synth
    from rsuper_amd.hip import lib, ops
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import init_distributed, wrap_ddp, train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    from rsuper_amd.training import losses_foundation as lf

    if args.force_ddp and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
    rank, local, world = init_distributed()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    local = local % torch.cuda.device_count()          # one rank per GPU; ranks only share a device in the gloo debugging mode
    torch.cuda.set_device(local)
    lib.require_device()
    dev = f'cuda:{local}'
    lf.SANITY_CHECKS = False             # the reference's per-step .sum()/isnan host syncs are checked once after the timed region
    classes = synth.PANTS_CLASSES
    B, S = args.batch, args.size

    torch.manual_seed(0)                 # identical random-init weights on every rank (DDP broadcasts anyway)
    net = UNet(1, args.base, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=args.dtype).to(dev)
    ema = make_ema(net)
    model = wrap_ddp(net, local) if (world > 1 or args.force_ddp) else net
    opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    kinds = (['mask', 'report'] * B)[:B] if args.report else ['mask'] * B
    bt = synth.batch(B, S, classes, kinds, seed=7 + rank, diam_range=(5.0, 40.0), max_tumors=3)
    batch = dict(image=torch.from_numpy(synth.image(B, S, seed=1234 + rank)).to(dev),
                 label=torch.from_numpy(bt['label']).to(dev), unk_channels=torch.from_numpy(bt['unk_channels']).to(dev),
                 mask=torch.from_numpy(bt['mask']).to(dev), volumes=torch.from_numpy(bt['volumes']).to(dev),
                 diameters=torch.from_numpy(bt['diameters']).to(dev))
    largs = argparse.Namespace(loss='ball_dice_both' if args.report else 'ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0,
                               report_volume_loss_basic=0.1 if args.report else 0.0, volume_loss_tolerance=0.2,
                               ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False,
                               stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        la, _ = train_step(model, ema, opt, batch, largs, classes, step)
        step += 1
    sync()
    timer = ops.KernelTimer() if rank == 0 else None
    ops.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        la, gn = train_step(model, ema, opt, batch, largs, classes, step)
        step += 1
    sync()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(la['overall'].detach())
    if not np.isfinite(loss_val):
        raise ValueError('loss is nan, propagating this can destroy the network weights, STOP!')

    if rank == 0:
        ksum = timer.summary()
        conv_ms_sum = sum(d['ms'] for d in ksum.values())
        conv_ms = timer.busy_ms            # union of the launch intervals: weight gradients overlap the data-gradient chain on a second stream
        conv_fl = sum(d['flops'] for d in ksum.values())
        fwd_fl = conv_stack_flops(args.base, S, B)
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out = {
            'metric': 'CT voxels/sec/node (96\u00b3 patch, bs=2/GPU)', 'value': world * B * S ** 3 * args.steps / dt, 'unit': 'voxels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'R-Super 3D UNet(base {args.base}, {len(classes)} classes, BasicBlock/IN) full training step '
                                   f'(fwd + {"seg+Volume+Ball" if args.report else "masked BCE + Dice"} loss + bwd + clip + AdamW + EMA), '
                                   f'{S}^3 patches, batch {B}/GPU (BASELINE.json configs[{2 if args.report else 1}])',
                       'global_batch': world * B, 'patch': S, 'parallelism': f'dp{world}', 'final_loss': loss_val},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': (CONV_TRAFFIC_BYTES_PER_STEP if (args.dtype == 'bf16' and args.base == 32 and B == 2 and S == 96 and not args.report) else None),
                         'traffic_unit': 'bytes per step over all conv MFMA launches (rocprofv3 PMC, profiles/r01_pmc_step.md)',
                         'sustained_mfma_peak_measured': 2110.0,
                         'kernel': 'conv3d MFMA kernels (igemm fwd + dgrad, wgrad): 3x the 43 3x3x3 convs',
                         'algorithmic_gflop_per_step': conv_fl / args.steps / 1e9, 'expected_gflop_per_step': 3 * fwd_fl / 1e9,
                         'conv_ms_per_step': conv_ms / args.steps, 'conv_ms_per_step_sum_of_launches': conv_ms_sum / args.steps,
                         'note': 'achieved = algorithmic FLOPs of all conv MFMA launches / union of their launch intervals (HIP events on the launch '
                                 'streams); per_kernel.avg_us are raw per-launch durations and include time shared with concurrently running kernels',
                         'step_level_frac': (3 * fwd_fl / (dt / args.steps)) / 1e12 / peak,
                         'per_kernel': {k: {'launches_per_step': d['launches'] / args.steps, 'avg_us': d['ms'] * 1e3 / d['launches'],
                                            'tflops': d['flops'] / (d['ms'] * 1e-3) / 1e12} for k, d in ksum.items()}},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args, classes)
    if world > 1 or args.force_ddp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: RCCL printf()s a version banner into the C stdio buffer, which would
        # otherwise be flushed at process exit, after this line
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
