"""world_size-2 worker on ONE MI355X (ranks share the device through RSUPER_DIST_BACKEND=gloo; RCCL refuses duplicate devices): the HIP training
path under wrap_ddp against the N > 1 parity definition of SURVEY.md section 8(e) -- rank r runs forward / calculate_loss / backward of the tiny UNet
(f32 parity mode) on ITS OWN batch, GradReducer exchanges the gradients behind the backward pass, and every parameter's gradient equals the mean of
the UNMODIFIED reference's two ranks (tests/golden/ddp2.npz, gen_golden_ddp.py)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import synth  # noqa: E402
from rsuper_amd.train_ddp import init_distributed, wrap_ddp  # noqa: E402
from rsuper_amd.model.dim3.unet import UNet  # noqa: E402
from rsuper_amd.training import losses_foundation as lf  # noqa: E402


def main():
    rank, local, world = init_distributed()
    assert world == 2
    torch.cuda.set_device(0)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ddp2.npz'))
    classes = synth.TINY_CLASSES
    net = UNet(1, 8, num_classes=len(classes), compute_dtype='f32')
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 3)
    if rank == 1:                                 # starts from other weights: wrap_ddp's rank-0 broadcast must repair it
        sd = {k: v + 0.5 for k, v in sd.items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to('cuda:0')
    model = wrap_ddp(net, 0)
    img, bt = synth.ddp_rank_batch(rank)
    args = argparse.Namespace(loss='ball_dice_both', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                              ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                              classification_branch=False, ema=True, ema_alpha=0.99)
    T = lambda a: torch.from_numpy(a).to('cuda:0')      # noqa: E731
    res = model(T(img))
    la = lf.calculate_loss(res, T(bt['label']), T(bt['unk_channels']), args, None, T(bt['mask']), T(bt['volumes']), T(bt['diameters']), classes)
    la['overall'].backward()
    red = getattr(net, '_rsuper_reducer', None)
    assert red is not None, 'wrap_ddp did not attach the GradReducer'
    red.finish()
    torch.cuda.synchronize()
    for k, v in la.items():
        assert abs(float(v.detach()) - float(g[f'r{rank}_{k}'])) <= 1e-4, (rank, k, float(v.detach()), float(g[f'r{rank}_{k}']))
    worst = 0.0
    for k, p in net.named_parameters():
        ref = g[f'mean_g_{k}_sub']
        sub, _ = synth.subsample(p.grad.cpu().numpy(), 2048)
        err = float(np.abs(sub - ref).max()) / max(float(g[f'mean_g_{k}_summary'][2]), 1e-12)
        worst = max(worst, err)
        assert err <= 2e-2, (rank, k, err)        # the bound of check_unet_tiny[f32] against the fp32 fixture (summation-order noise of the tiny network)
    dist.barrier()
    if rank == 0:
        print(f'DDP_FIXTURE_GPU_OK worst {worst:.2e}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
