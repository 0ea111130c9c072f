#!/usr/bin/env python3
"""Where in the backward pass is each gradient bucket handed to the collective?  (evidence of the exchange / backward overlap that a
single-GPU box can give.)  `bench.py --force-ddp`'s configuration -- the full UNet under wrap_ddp on a 1-rank RCCL group, i.e. the GradReducer path of the
driver's N > 1 runs -- with a HIP event recorded on the launching stream at every bucket's all-reduce call and at the start / end of backward.  On one
rank RCCL launches NO device kernel for an all-reduce (rocprofv3 kernel trace of the same run: 0 collective kernels, profiles/r04_ddp_overlap.txt), so
the device timeline cannot show the collective itself; what it shows is the point of the backward at which each bucket becomes available to the
interconnect and how much compute is still queued behind it -- the time a real exchange has to hide in.
Usage: python tools/ddp_bucket_timeline.py [steps]"""
import os, sys
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29591')
import synth
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.train_ddp import wrap_ddp, make_ema, train_step
from rsuper_amd.training.utils import FusedAdamWEMA
from rsuper_amd.training import losses_foundation as lf
from rsuper_amd import reducer as R
import bench

dist.init_process_group(backend='nccl', rank=0, world_size=1)
torch.cuda.set_device(0)
lf.SANITY_CHECKS = False
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
classes = synth.PANTS_CLASSES
torch.manual_seed(0)
net = UNet(1, 32, num_classes=26, compute_dtype='bf16').to('cuda')
ema = make_ema(net)
model = wrap_ddp(net, 0)
red = net._rsuper_reducer
opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
bt = synth.batch(2, 96, classes, ['mask', 'mask'], seed=7)
batch = dict(image=torch.from_numpy(synth.image(2, 96, seed=1234)).cuda(), **{k: torch.from_numpy(v).cuda() for k, v in bt.items()})
args = bench.loss_args(False)

marks = []
orig_launch = R.GradReducer._launch
def launch(self, b):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((self.buckets.index(b), b.flat.numel() * 4 / 2 ** 20, ev))
    return orig_launch(self, b)
R.GradReducer._launch = launch
orig_bwd = torch.Tensor.backward
tl = {}
def bwd(self, *a, **k):
    tl['b0'] = torch.cuda.Event(enable_timing=True); tl['b0'].record()
    r = orig_bwd(self, *a, **k)
    tl['b1'] = torch.cuda.Event(enable_timing=True); tl['b1'].record()
    return r
torch.Tensor.backward = bwd

print(f'# {len(red.buckets)} buckets (MB): ' + ', '.join(f'{b.flat.numel() * 4 / 2 ** 20:.1f}' for b in red.buckets) + f'; {sum(b.flat.numel() for b in red.buckets) * 4 / 2 ** 20:.1f} MB of f32 gradients per step')
for s in range(steps):
    del marks[:]
    tl['s0'] = torch.cuda.Event(enable_timing=True); tl['s0'].record()
    train_step(model, ema, opt, batch, args, classes, s)
    tl['s1'] = torch.cuda.Event(enable_timing=True); tl['s1'].record()
    torch.cuda.synchronize()
    if s < 2:
        continue
    bw = tl['b0'].elapsed_time(tl['b1'])
    line = f'step {s}: {tl["s0"].elapsed_time(tl["s1"]):.2f} ms, backward {bw:.2f} ms (device time between the first and the last backward launch) |'
    for i, mb, ev in marks:
        t = tl['b0'].elapsed_time(ev)
        line += f' bucket {i} ({mb:.1f} MB) handed over at {t:.2f} ms = {100 * t / bw:.0f} % of backward, {bw - t:.2f} ms of backward still queued |'
    print(line)
dist.destroy_process_group()
