#!/usr/bin/env python3
"""Sliding-window inference throughput: full UNet (base 32, 26 classes), 96^3 window, half-window overlap,
volume 192^3 resident in HBM.  Prints one JSON line.   python tools/bench_infer.py [--dtype bf16|f32] [--size 192]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--size', type=int, default=192)
    ap.add_argument('--window', type=int, default=96)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--window-batch', type=int, default=8)
    a = ap.parse_args()
    import synth
    from rsuper_amd.hip import lib
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.inference import inference_sliding_window
    lib.require_device()
    classes = synth.PANTS_CLASSES
    torch.manual_seed(0)
    net = UNet(1, 32, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=a.dtype).to('cuda')
    img = torch.randn((1, 1, a.size, a.size, a.size), device='cuda').clamp_(-3, 3)
    args = argparse.Namespace(window_size=[a.window] * 3, classes=len(classes))
    inference_sliding_window(net, img, args, to_cpu=False, window_batch=a.window_batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = inference_sliding_window(net, img, args, to_cpu=False, window_batch=a.window_batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    nwin = (a.size // (a.window // 2)) ** 3
    print(json.dumps({'metric': 'sliding-window inference voxels/s (output voxels)', 'value': a.size ** 3 / dt, 'unit': 'voxels/s',
                      'ms_per_volume': dt * 1e3, 'windows': nwin, 'ms_per_window': dt * 1e3 / nwin, 'dtype': a.dtype,
                      'config': {'workload': f'UNet(base 32, {len(classes)} classes) forward, {a.window}^3 windows at half-window stride over a {a.size}^3 volume, '
                                             f'device-side sigmoid accumulation + normalisation'},
                      'checks': {'min': float(out.min()), 'max': float(out.max())}}))


if __name__ == '__main__':
    main()
