import sys, os, argparse
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, synth
import torch.distributed as dist
from rsuper_amd.hip import ops
from rsuper_amd.model.dim3.unet import UNet
from rsuper_amd.train_ddp import train_step, wrap_ddp, make_ema
from rsuper_amd.training.utils import FusedAdamWEMA
DEV='cuda'
classes = synth.TINY_CLASSES
B, S = 2, 32
bt = synth.batch(B, S, classes, ['mask', 'mask'], seed=5)
batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
batch['image'] = torch.from_numpy(synth.image(B, S, seed=3)).to(DEV)
la = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                        ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                        classification_branch=False, ema=True, ema_alpha=0.99)
def run(wrapped, steps=1):
    torch.manual_seed(0)
    net = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='f32').to(DEV)
    ema = make_ema(net)
    model = wrap_ddp(net, 0) if wrapped else net
    opt = FusedAdamWEMA(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    gs = None
    for step in range(steps):
        l, gn = train_step(model, ema, opt, batch, la, classes, step)
        print('wrapped' if wrapped else 'plain', step, repr(float(gn)), repr(float(l['overall'])), sum(int(p.grad.data_ptr() % 16 != 0) for p in net.parameters()))
        if step == 0: gs = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    if wrapped: net._rsuper_reducer.remove()
    return gs, {k: p.detach().clone() for k, p in net.named_parameters()}
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
dist.init_process_group(backend='nccl', rank=0, world_size=1)
ga, a = run(True, 2)
dist.destroy_process_group(); ops.GRAD_DEST = None
gb, b = run(False, 2)
for k in a:
    dg = (ga[k]-gb[k]).abs().max().item(); dp = (a[k]-b[k]).abs().max().item()
    if 0: print(k, 'grad diff', dg, 'param diff', dp, 'gmax', gb[k].abs().max().item())
print('done')
