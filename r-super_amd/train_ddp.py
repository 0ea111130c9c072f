"""Training step and DDP worker mirroring rsuper_train/train_ddp.py (train_epoch :235-389, main_worker :593-691).

`train_step` is the hot loop body of train_epoch (:308-357):
    zero_grad -> net(img) -> calculate_loss -> backward -> clip_grad_norm_(1.0) -> AdamW.step -> EMA
with clip + AdamW + EMA fused into one HBM pass.  One process per GPU; gradients are averaged with RCCL
(`torch.distributed` backend 'nccl') by torch's DistributedDataParallel, whose bucketed all-reduce overlaps the
backward kernels because every block's weight gradients are separate autograd leaves.
"""
import copy
import os

import torch
import torch.distributed as dist

from .training import losses_foundation as lf
from .training.utils import FusedAdamWEMA, ema_alpha_for_step, get_optimizer, exp_lr_scheduler_with_warmup, unwrap_model_checkpoint


def make_ema(net):
    """EMA copy as in init_network (train_ddp.py:569-571): same architecture, gradients disabled."""
    ema = copy.deepcopy(net.module if hasattr(net, 'module') else net)
    for p in ema.parameters():
        p.requires_grad_(False)
    return ema


def train_step(net, ema_net, optimizer, batch, args, classes, step, matcher=None):
    """One iteration of train_epoch (:308-357).  batch: dict with the reference keys image, label, unk_channels,
    volumes, mask, diameters[, weights] already on the device.  Returns the loss dict (device tensors) and the
    pre-clip gradient norm."""
    img = batch['image']
    optimizer.zero_grad(set_to_none=True)
    result = net(img)
    loss_all = lf.calculate_loss(model_output=result, label=batch['label'], unk_voxels=batch.get('unk_channels'), args=args,
                                 matcher=matcher, chosen_segment_mask=batch.get('mask'),
                                 tumor_volumes_report=batch.get('volumes'), tumor_diameters=batch.get('diameters'),
                                 classes=classes, input_tensor=img, class_weights=batch.get('weights'))
    loss_all['overall'].backward()
    reducer = getattr(net, '_rsuper_reducer', None)
    if reducer is not None:
        reducer.finish()                 # gradients are now the mean over ranks (what DDP's reducer leaves in p.grad)
    gnorm = None
    if isinstance(optimizer, FusedAdamWEMA):
        ema_params = list(ema_net.parameters()) if (ema_net is not None and getattr(args, 'ema', True)) else None
        gnorm = optimizer.fused_step(max_norm=1.0, ema_params=ema_params,
                                     ema_alpha=ema_alpha_for_step(getattr(args, 'ema_alpha', 0.99), step))
    else:
        gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        optimizer.step()
        if ema_net is not None and getattr(args, 'ema', True):
            from .training.utils import update_ema_variables
            update_ema_variables(net, ema_net, args.ema_alpha, step)
    return loss_all, gnorm


def save_checkpoint(path, epoch, net, ema_net, optimizer, args=None):
    """Checkpoint dict of train_ddp.py:184-189 (plain state_dicts; reference loaders accept them)."""
    sd, esd = unwrap_model_checkpoint(net, ema_net, args)
    torch.save({'epoch': epoch + 1, 'model_state_dict': sd, 'ema_model_state_dict': esd,
                'optimizer_state_dict': optimizer.state_dict()}, path)


def load_checkpoint(path, net, ema_net=None, optimizer=None, map_location='cpu'):
    """resume_load_* (rsuper_train/utils.py:41-62): accepts state_dicts or whole modules under the model keys."""
    ck = torch.load(path, map_location=map_location, weights_only=False)

    def sd_of(o):
        return o.state_dict() if isinstance(o, torch.nn.Module) else o
    (net.module if hasattr(net, 'module') else net).load_state_dict(sd_of(ck['model_state_dict']), strict=False)
    if ema_net is not None and ck.get('ema_model_state_dict') is not None:
        (ema_net.module if hasattr(ema_net, 'module') else ema_net).load_state_dict(sd_of(ck['ema_model_state_dict']), strict=False)
    if optimizer is not None and 'optimizer_state_dict' in ck:
        optimizer.load_state_dict(ck['optimizer_state_dict'])
    return ck.get('epoch', 0)


def init_distributed(backend=None):
    """One process per GPU (train_ddp.py:623-632); RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment.
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 0, 1
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    if backend is None:
        # RSUPER_DIST_BACKEND=gloo: debugging aid -- several ranks may then share one GPU (RCCL refuses duplicate devices),
        # which exercises the multi-rank control flow of bench.py / GradReducer on a single-GPU box
        backend = os.environ.get('RSUPER_DIST_BACKEND', 'nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def wrap_ddp(net, local_rank, bucket_cap_mb=25):
    """Data-parallel wrapper of main_worker (train_ddp.py:663, DistributedDataParallel(net, device_ids=[idx],
    find_unused_parameters=False)).  On the GPU the default is rsuper_amd.reducer.GradReducer attached to the bare module
    (same mathematics: rank-0 parameter broadcast + mean of gradients; flat buckets the HIP kernels write into, one async
    RCCL all-reduce per bucket from the weight-gradient stream) -- `train_step` calls its finish() after backward.
    RSUPER_REDUCER=0 (or a CPU module) selects torch's DistributedDataParallel with gradient_as_bucket_view."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if next(net.parameters()).is_cuda and os.environ.get('RSUPER_REDUCER', '1') == '1':
        from .reducer import GradReducer
        net._rsuper_reducer = GradReducer(net, bucket_mb=int(os.environ.get('RSUPER_DDP_BUCKET_MB', 48)))
        return net
    if next(net.parameters()).is_cuda:
        bucket_cap_mb = int(os.environ.get('RSUPER_DDP_BUCKET_MB', bucket_cap_mb))
        return DDP(net, device_ids=[local_rank], find_unused_parameters=False, gradient_as_bucket_view=True,
                   bucket_cap_mb=bucket_cap_mb, broadcast_buffers=False,
                   static_graph=os.environ.get('RSUPER_DDP_STATIC', '0') == '1')
    return DDP(net, find_unused_parameters=False)


def shard_indices(chunk, rank, world_size):
    """Round-robin rank sharding of an epoch chunk (training/dataset/dim3/sampler.py:132)."""
    return chunk[rank::world_size]
