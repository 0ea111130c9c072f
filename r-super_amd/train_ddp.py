"""Training step and DDP worker mirroring rsuper_train/train_ddp.py (train_epoch :235-389, main_worker :593-691).

`train_step` is the hot loop body of train_epoch (:308-357):
    zero_grad -> net(img) -> calculate_loss -> backward -> clip_grad_norm_(1.0) -> AdamW.step -> EMA
with clip + AdamW + EMA fused into one HBM pass.  One process per GPU; gradients are averaged with RCCL
(`torch.distributed` backend 'nccl'): by `rsuper_amd.reducer.GradReducer` (default of `wrap_ddp`: flat buckets the weight-gradient
kernels write into, one asynchronous all-reduce per bucket overlapping the data-gradient chain), by `GraphedNetwork.exchange_gradients`
when the network is replayed from hipGraphs, or by torch's DistributedDataParallel (`RSUPER_REDUCER=0`, CPU modules).
"""
import argparse
import copy
import logging
import os
import random
import sys
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .training import losses_foundation as lf
from .training.dataset.packed import ingest_packed_batch
from .training.utils import FusedAdamWEMA, ema_alpha_for_step, get_optimizer, exp_lr_scheduler_with_warmup, unwrap_model_checkpoint


def make_ema(net):
    """EMA copy as in init_network (train_ddp.py:569-571): same architecture, gradients disabled."""
    ema = copy.deepcopy(net.module if hasattr(net, 'module') else net)
    for p in ema.parameters():
        p.requires_grad_(False)
    return ema


class StepGuard:
    """The reference's per-step guards -- input range asserts (train_ddp.py:311-313), mask / unknown / volume consistency (losses_foundation.py:864-869),
    NaN loss (:1070-1071) -- without their device->host synchronisations (six per step in the reference; +0.6 ms per step here when done the same way).
    Every condition ORs a flag on the device; `end_step()` snapshots the flags into pinned memory behind an event and clears them; `poll()` -- called after
    the NEXT step has been queued, so the wait hides behind it -- raises the reference's exception for the step before.  Weights stay intact meanwhile:
    a NaN loss gives a non-finite gradient norm, for which the fused optimiser skips the update (csrc/optim.hip).  `poll(final=True)` drains at the end
    of an epoch."""
    MESSAGES = ((AssertionError, 'Input is nan'), (AssertionError, 'Input is bigger than 100'), (AssertionError, 'Input is smaller than -100'),
                (ValueError, 'unk_voxels should not be all zeros if chosen_segment_mask is not all zeros'),
                (ValueError, 'tumor_volumes_report should not be all zeros if chosen_segment_mask is not all zeros'),
                (ValueError, 'loss is nan, propagating this can destroy the network weights, STOP!'))

    def __init__(self, device):
        self.flags = torch.zeros(8, dtype=torch.int32, device=device)
        self.host = [torch.zeros(8, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.events = [None, None]
        self.n = 0

    def check_input(self, img):
        from .hip import lib as _l, ops as _ops
        x = img if (img.dtype == torch.float32 and img.is_contiguous()) else img.float().contiguous()
        _l.check(_l.lib().rsuper_guard_range(x.data_ptr(), x.numel(), -100.0, 100.0, self.flags.data_ptr(), _ops._stream()), 'guard_range')

    def consistency(self, m_any, u_any, volumes):
        """losses_foundation.py:864-869 on the per-sample any-bytes of the segment mask / unknown map and the (B, T) report volumes -> flags 3, 4."""
        from .hip import lib as _l, ops as _ops
        v = volumes if (volumes.dtype == torch.float32 and volumes.is_contiguous()) else volumes.float().contiguous()
        B = int(m_any.numel())
        _l.check(_l.lib().rsuper_guard_consistency(m_any.data_ptr(), u_any.data_ptr(), v.data_ptr(), B, int(v.numel() // B), self.flags.data_ptr() + 12,
                                                   _ops._stream()), 'guard_consistency')

    def nan(self, x):
        """flag 5 |= any NaN in the f32 tensor x (the loss)."""
        from .hip import lib as _l, ops as _ops
        x = x.reshape(-1).float().contiguous()
        _l.check(_l.lib().rsuper_guard_range(x.data_ptr(), x.numel(), float('-inf'), float('inf'), self.flags.data_ptr() + 20, _ops._stream()), 'guard_nan')

    def end_step(self):
        k = self.n & 1
        self.host[k].copy_(self.flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        self.flags.zero_()
        self.n += 1

    def _check(self, k):
        ev = self.events[k]
        if ev is None:
            return
        ev.synchronize()
        self.events[k] = None
        h = self.host[k]
        for i, (exc, msg) in enumerate(self.MESSAGES):
            if int(h[i]) != 0:
                raise exc(msg + ' (raised one step late: device-side guard; the update of a step with a non-finite gradient norm was skipped)')

    def poll(self, final=False):
        """After end_step() of step s: checks step s - 1 (final: also step s)."""
        self._check(self.n & 1)              # the older snapshot
        if final:
            self._check((self.n + 1) & 1)


PREFETCH_SUPERVISION = os.environ.get('RSUPER_PREFETCH_SUPERVISION', '1') == '1'      # =0: the report losses read their host inputs inside calculate_loss (A/B)


def train_step(net, ema_net, optimizer, batch, args, classes, step, matcher=None):
    """One iteration of train_epoch (:308-357).  batch: dict with the reference keys image, label, unk_channels,
    volumes, mask, diameters[, weights] already on the device.  Returns the loss dict (device tensors) and the
    pre-clip gradient norm."""
    img = batch['image']
    optimizer.zero_grad(set_to_none=True)
    # the batch-only inputs of the report losses (dilated segment masks, three small host reads) go into the queue AHEAD of the forward pass
    pre = lf.prepare_report_supervision(batch['label'], batch.get('unk_channels'), batch.get('mask'), batch.get('volumes'), batch.get('diameters'),
                                        classes, args) if PREFETCH_SUPERVISION else None
    result = net(img)
    loss_all = lf.calculate_loss(model_output=result, label=batch['label'], unk_voxels=batch.get('unk_channels'), args=args,
                                 matcher=matcher, chosen_segment_mask=batch.get('mask'),
                                 tumor_volumes_report=batch.get('volumes'), tumor_diameters=batch.get('diameters'),
                                 classes=classes, input_tensor=img, class_weights=batch.get('weights'), pre=pre)
    loss_all['overall'].backward()
    reducer = getattr(net, '_rsuper_reducer', None)
    if reducer is not None:
        reducer.finish()                 # gradients are now the mean over ranks (what DDP's reducer leaves in p.grad)
    gnorm = None
    if isinstance(optimizer, FusedAdamWEMA):
        ema_params = None
        if ema_net is not None and getattr(args, 'ema', True):
            # the module-tree walk of .parameters() costs ~1 ms per step on MedFormer's 300 modules: keep the list on the module
            ema_params = ema_net.__dict__.get('_rsuper_param_list')
            if ema_params is None:
                ema_params = ema_net.__dict__['_rsuper_param_list'] = list(ema_net.parameters())
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
        net._rsuper_reducer = GradReducer(net, bucket_mb=int(os.environ.get('RSUPER_DDP_BUCKET_MB', 48)),
                                          wire_dtype=torch.bfloat16 if os.environ.get('RSUPER_DDP_BF16', '0') == '1' else None)
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


# ------------------------------------------------------------------------------------------------ epoch loop / driver
class AverageMeter:
    """utils.AverageMeter of the reference (rsuper_train/utils.py): running value / average of one logged quantity."""

    def __init__(self, name, fmt=':f'):
        self.name, self.fmt = name, fmt
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ('{name} {val' + self.fmt + '} ({avg' + self.fmt + '})').format(**self.__dict__)


class ProgressMeter:
    def __init__(self, num_batches, meters, prefix=''):
        n = len(str(num_batches // 1))
        self.fmt = '[{:' + str(n) + 'd}/' + ('{:' + str(n) + 'd}').format(num_batches) + ']'
        self.meters, self.prefix = meters, prefix

    def display(self, batch):
        logging.info('\t'.join([self.prefix + self.fmt.format(batch)] + [str(m) for m in self.meters]))


def is_master(args):
    return getattr(args, 'rank', 0) % max(getattr(args, 'ngpus_per_node', 1), 1) == 0 if getattr(args, 'distributed', False) else True


def train_epoch(trainLoader, net, ema_net, optimizer, epoch, writer, scaler, args, matcher=None):
    """train_epoch (train_ddp.py:235-389): same signature and control flow -- batch dictionary keys (:247-262), host->device
    copies, the input range asserts (:311-313), one `train_step` per batch (zero_grad -> forward -> calculate_loss ->
    backward -> clip -> optimizer -> EMA), loss meters over every key calculate_loss returns, progress lines every
    args.print_freq, the 3D epoch length check (`iter_num_per_epoch > args.iter_per_epoch`, :380-383: a loader with more
    batches would run iter_per_epoch + 1 iterations; the ChunkedSampler of train_net serves exactly iter_per_epoch) and the
    per-epoch TensorBoard scalars."""
    if getattr(args, 'amp', False):
        raise ValueError('MedFormer seems unstable with amp, please use float32 precision')        # train_ddp.py:316
    net.train()
    start = time.time()
    loss_meters = OrderedDict()
    progress = None
    iter_num_per_epoch = 0
    dev = next(net.parameters()).device
    classes = trainLoader.dataset.classes
    packed = getattr(trainLoader.dataset, 'packed', False)
    # rsuper_amd extension (--hip_graph): replay the step from a hipGraph where that is possible -- segmentation-only supervision, one process,
    # the fused optimiser (rsuper_amd/graph.py); the stepper lives on the optimiser so that it survives across epochs
    # With report supervision (host-synchronous ball search in the loss) the network's forward and backward are replayed from two graphs
    # around the eager loss and optimiser (graph.GraphedNetwork); a batch of another shape (last batch of an epoch) takes the eager path.
    stepper, fwd_net = None, net
    if getattr(args, 'hip_graph', False) and getattr(net, '_rsuper_reducer', None) is None and not hasattr(net, 'module'):
        if float(getattr(args, 'report_volume_loss_basic', 0.0)) == 0 and isinstance(optimizer, FusedAdamWEMA) \
                and not getattr(args, 'distributed', False):
            from .graph import GraphedTrainStep
            stepper = getattr(optimizer, '_graphed_step', None)
            if stepper is None or stepper.net is not net:
                stepper = optimizer._graphed_step = GraphedTrainStep(net, ema_net, optimizer, args, classes)
        else:
            from .graph import GraphedNetwork
            fwd_net = getattr(optimizer, '_graphed_net', None)
            if fwd_net is None or fwd_net.net is not net:
                fwd_net = optimizer._graphed_net = GraphedNetwork(net)
    # the reference's guards as device flags (StepGuard); RSUPER_ASYNC_GUARDS=0: synchronise where the reference does.  Replayed steps keep the host-side
    # NaN test below (their loss code is not re-run)
    # Only with the fused optimiser: its kernel skips the update of a step whose gradient norm is not finite (csrc/optim.hip), which is what makes reading the
    # flags one step late safe.  Any other optimiser keeps the reference's synchronous checks (the NaN loss raises BEFORE backward, losses_foundation.py:1070).
    guard = StepGuard(dev) if (lf.SANITY_CHECKS and stepper is None and dev.type == 'cuda' and isinstance(optimizer, FusedAdamWEMA)
                               and os.environ.get('RSUPER_ASYNC_GUARDS', '1') == '1') else None
    lf.GUARD = guard
    try:
        return _train_epoch_loop(trainLoader, net, ema_net, optimizer, epoch, writer, args, matcher, guard, stepper, fwd_net, dev, classes, packed, start)
    finally:
        lf.GUARD = None


def _train_epoch_loop(trainLoader, net, ema_net, optimizer, epoch, writer, args, matcher, guard, stepper, fwd_net, dev, classes, packed, start):
    loss_meters = OrderedDict()
    progress = None
    iter_num_per_epoch = 0
    for i, inputs in enumerate(trainLoader):
        batch = dict(image=inputs['image'], label=inputs['label'], unk_channels=inputs['unk_channels'],
                     volumes=inputs['volumes'].float(), mask=inputs['mask'], diameters=inputs['diameters'].float())
        if 'weights' in inputs:
            batch['weights'] = inputs['weights'].float()
        if packed:      # bit-packed label volumes cross PCIe as stored and are inflated on the device (dataset/packed.py)
            # the three volumes never leave their packed form: the loss kernels read the label bits, the report losses inflate the lesion planes they index
            # and take the unknown map's plane flags from the packed bytes (calculate_loss) -- with and without report supervision
            batch = ingest_packed_batch(batch, len(classes), dev, keep_packed=True)
        else:
            batch = {k: v.to(dev, non_blocking=True) for k, v in batch.items()}
        img = batch['image']
        step = i + epoch * len(trainLoader)                      # global steps (:306)
        if lf.SANITY_CHECKS and guard is not None:               # the same three asserts (:311-313) as device flags, raised by guard.poll()
            guard.check_input(img)
        elif lf.SANITY_CHECKS:
            assert not torch.isnan(img).any(), 'Input is nan'
            assert torch.max(img) <= 100, f'Input is bigger than 100: {torch.max(img)}'
            assert torch.min(img) >= -100, f'Input is smaller than -100: {torch.min(img)}'
        if stepper is not None:
            loss_all, _ = stepper(batch, step)
        else:
            # another batch shape: eager module -- unless the graphed network also carries the gradient exchange (then it refuses loudly)
            use = fwd_net if (fwd_net is net or fwd_net.accepts(img) or fwd_net.exchange) else net
            loss_all, _ = train_step(use, ema_net, optimizer, batch, args, classes, step, matcher=matcher)
        if guard is not None:
            guard.end_step()
        if len(loss_meters) == 0:
            loss_meters = OrderedDict((k, AverageMeter(k, ':6.4f')) for k in loss_all.keys())
            loss_meters['Elapsed Time'] = AverageMeter('Elapsed Time', ':6.2f')
        for k, v in loss_all.items():
            val = v.item()
            if k == 'overall' and val != val:            # the NaN guard of calculate_loss (:1070-1071) also for replayed steps
                # eager steps raise inside calculate_loss, before backward; a REPLAYED step has already run clip / AdamW / EMA with the
                # NaN gradients when the host reads the loss here: net and ema_net are then invalid and must be reloaded from the last
                # checkpoint (train_net keeps `latest`); in a multi-rank run stop all ranks (the others block in the next collective)
                raise ValueError('loss is nan, propagating this can destroy the network weights, STOP!'
                                 + (' (hipGraph replay: the weights of this step are already updated -- resume from the last checkpoint)'
                                    if stepper is not None else ''))
            loss_meters[k].update(val, img.shape[0])
        if guard is not None:
            guard.poll(final=True)               # the meters above have synchronised on this step already: its flags are there, raise now
        loss_meters['Elapsed Time'].update(time.time() - start, n=1)
        if progress is None:
            progress = ProgressMeter(len(trainLoader) if args.dimension == '2d' else args.iter_per_epoch, list(loss_meters.values()),
                                     prefix=f"{getattr(args, 'unique_name', 'test')} epoch: [{epoch + 1}]")
        if i % args.print_freq == 0:
            progress.display(i)
        if args.dimension == '3d':
            iter_num_per_epoch += 1
            if iter_num_per_epoch > args.iter_per_epoch:
                break
    if is_master(args) and writer is not None:
        for key, meter in loss_meters.items():
            writer.add_scalar(f'Train/{key}', meter.avg, epoch + 1)
    return loss_meters


def merge_config(args, config):
    """The YAML merge of get_parser (train_ddp.py:491-502): a config key only fills an attribute the command line does not
    define at all (`if not hasattr(args, key)`), so every argparse option -- set or defaulted -- wins over the file."""
    for key, value in config.items():
        if not hasattr(args, key):
            setattr(args, key, value)
    return args


def get_parser(argv=None, config_root=None):
    """The command line of train_ddp.py:392-548 restricted to the options the accelerated path reads (same names, types and
    defaults), the YAML merge and the override rules that follow it (:504-546)."""
    import yaml
    parser = argparse.ArgumentParser(description='R-Super 3D segmentation training on MI355X (rsuper_amd)')
    parser.add_argument('--dataset', type=str, default='abdomenatlas_ufo')
    parser.add_argument('--reports', default=None)
    parser.add_argument('--model', type=str, default='unet')
    parser.add_argument('--dimension', type=str, default='3d')
    parser.add_argument('--pretrain', action='store_true')
    parser.add_argument('--amp', action='store_true')
    parser.add_argument('--batch_size', default=2, type=int, help='GLOBAL batch size (divided by the GPUs of the node, :632)')
    parser.add_argument('--resume', action='store_true')
    parser.add_argument('--cp_path', type=str, default='./exp/')
    parser.add_argument('--log_path', type=str, default='./log/')
    parser.add_argument('--unique_name', type=str, default='test')
    parser.add_argument('--workers', type=int, default=None)
    parser.add_argument('--data_root', type=str, default=None)
    parser.add_argument('--UFO_root', type=str, default=None)
    parser.add_argument('--world_size', type=int, default=1)
    parser.add_argument('--rank', type=int, default=0)
    parser.add_argument('--dist_url', type=str, default='tcp://127.0.0.1:8001')
    parser.add_argument('--dist_backend', type=str, default='nccl')
    parser.add_argument('--report_volume_loss_basic', type=float, default=1)
    parser.add_argument('--seg_loss', type=float, default=1)
    parser.add_argument('--warmup', type=int, default=5)
    parser.add_argument('--loss', type=str, default='ball_dice_last')
    parser.add_argument('--classification_branch', action='store_true')
    parser.add_argument('--multi_ch_tumor', action='store_true')
    parser.add_argument('--model_genesis_pretrain', action='store_true')
    parser.add_argument('--update_output_layer', action='store_true', help='rebuild the output heads for the class list of the dataset, keeping the kernels of the classes in --old_classes (train_ddp.py:437)')
    parser.add_argument('--old_classes', type=str, default=None, help='yaml file with the class list of the checkpoint (sorted on load, train_ddp.py:438,647-652)')
    parser.add_argument('--no_mask', action='store_true', help='report-only training; with --update_output_layer new classes start from the pancreatic_lesion kernels (train_ddp.py:454,577)')
    parser.add_argument('--pretrained', type=str, default=None, help='pretrained model path (train_ddp.py:431)')
    parser.add_argument('--clip_pretrain', action='store_true')
    parser.add_argument('--epochs', type=int, default=None)
    parser.add_argument('--classes_number', type=int, default=None)
    parser.add_argument('--ball_bce_weight', type=float, default=1)
    parser.add_argument('--ball_dice_weight', type=float, default=1)
    parser.add_argument('--stardard_ce_ball', action='store_true')
    parser.add_argument('--lr', type=float, default=0.0006)
    parser.add_argument('--ball_volume_margin', type=float, default=0.2)
    parser.add_argument('--volume_loss_tolerance', type=float, default=0.2)
    parser.add_argument('--crop_size', default=None, type=int)
    parser.add_argument('--load_augmented', action='store_true', help='Loads pre-saved crops for training (:414)')
    parser.add_argument('--save_destination', type=str, default=None, help='directory of the pre-saved crops (:415)')
    parser.add_argument('--hip_graph', action='store_true', help='rsuper_amd extension: replay the training step from a hipGraph (segmentation-only supervision) or the network forward / backward from two graphs around the eager loss (report supervision)')
    parser.add_argument('--synthetic', type=int, default=0, help='rsuper_amd extension: train on N synthetic samples (no dataset on disk)')
    args = parser.parse_args(argv)

    reports, dr, epochs, ufo_root, w, lr, classes_number = args.reports, args.data_root, args.epochs, args.UFO_root, args.workers, args.lr, args.classes_number
    root = config_root or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config')
    config_path = os.path.join(root, args.dataset, f'{args.model}_{args.dimension}.yaml')
    if not os.path.exists(config_path):
        raise ValueError("The specified configuration doesn't exist: %s" % config_path)
    with open(config_path, 'r') as f:
        config = yaml.load(f, Loader=yaml.SafeLoader)
    merge_config(args, config)
    # overrides after the merge (:513-529): options that exist on BOTH sides under different names, or that argparse always defines
    if w is not None:
        args.num_workers = w
    if dr is not None:
        args.data_root = dr
    if epochs is not None:
        args.epochs = epochs
    if ufo_root is not None:
        args.UFO_root = ufo_root
    if classes_number is not None:
        args.classes = classes_number
    if lr is not None:
        args.base_lr = lr
    if reports is not None:
        args.reports = reports
    if args.crop_size is not None:
        args.training_size = [args.crop_size] * 3
    for k, v in dict(num_workers=0, start_epoch=0, aug_device='cpu', val_freq=10 ** 9).items():
        if not hasattr(args, k):
            setattr(args, k, v)
    args.batch_size_global = args.batch_size
    return args


def train_net(net, trainset, testset, args, ema_net=None, fold_idx=0, writer=None):
    """Training half of train_net (train_ddp.py:65-232): ChunkedSampler + DataLoader (:105-123), optimizer (:141), per epoch
    the sampler reshuffle, the exponential warm-up / polynomial LR schedule (:170) and train_epoch, the `latest` checkpoint of
    every epoch and the numbered one every 25 (:181-200).  Validation (:203-230) is outside the hot path."""
    from torch.utils import data
    from .training.dataset import ChunkedSampler
    leng = len(trainset.img_list) if hasattr(trainset, 'img_list') else len(trainset)
    distributed = getattr(args, 'distributed', False)
    ngpus = getattr(args, 'ngpus_per_node', 1)
    sampler = ChunkedSampler(dataset_size=leng, samples_per_epoch=args.iter_per_epoch * args.batch_size * ngpus, shuffle=True, seed=42,
                             rank=dist.get_rank() if distributed else 0, world_size=dist.get_world_size() if distributed else 1)
    loader = data.DataLoader(trainset, batch_size=args.batch_size, shuffle=False, sampler=sampler, pin_memory=(args.aug_device != 'gpu'),
                             num_workers=args.num_workers, persistent_workers=(args.num_workers > 0))
    optimizer = get_optimizer(args, net)
    cp_dir = os.path.join(args.cp_path, args.dataset, args.unique_name)
    if getattr(args, 'resume', False):
        args.start_epoch = load_checkpoint(os.path.join(cp_dir, f'fold_{fold_idx}_latest.pth'), net, ema_net, optimizer)
    if args.epochs is None:
        raise ValueError('--epochs is required: argparse defines the attribute, so the YAML value never fills it (train_ddp.py:491-502)')
    history = []
    for epoch in range(args.start_epoch, args.epochs):
        sampler.set_epoch(epoch)
        lr = exp_lr_scheduler_with_warmup(optimizer, epoch=epoch, warmup_epoch=args.warmup, max_epoch=args.epochs)
        logging.info(f'Starting epoch {epoch + 1}/{args.epochs}, lr {lr:.4e}')
        meters = train_epoch(loader, net, ema_net, optimizer, epoch, writer, None, args, matcher=None)
        history.append({k: m.avg for k, m in meters.items()})
        if is_master(args):
            os.makedirs(cp_dir, exist_ok=True)
            save_checkpoint(os.path.join(cp_dir, f'fold_{fold_idx}_latest.pth'), epoch, net, ema_net, optimizer, args)
            if (epoch + 1) % 25 == 0:
                save_checkpoint(os.path.join(cp_dir, f'fold_{fold_idx}_epoch_{epoch + 1}.pth'), epoch, net, ema_net, optimizer, args)
    return history


def main_worker(proc_idx, ngpus_per_node, fold_idx, args, result_dict=None, trainset=None, testset=None):
    """main_worker (train_ddp.py:593-691): per-process seeding (:595-604), one process per GPU with the global batch divided
    by the GPUs of the node (:632), model + EMA construction (init_network :549-590), the data-parallel wrapper (:663) and
    train_net.  Ranks come from the launcher environment (torchrun / bench.py) or from proc_idx under mp.spawn."""
    from .model.utils import get_model
    if getattr(args, 'reproduce_seed', None) is not None:
        random.seed(args.reproduce_seed); np.random.seed(args.reproduce_seed); torch.manual_seed(args.reproduce_seed)
    args.proc_idx, args.ngpus_per_node = proc_idx, ngpus_per_node
    args.distributed = int(os.environ.get('WORLD_SIZE', getattr(args, 'world_size', 1))) > 1
    if args.distributed:
        os.environ.setdefault('RANK', str(args.rank * ngpus_per_node + proc_idx))
        os.environ.setdefault('LOCAL_RANK', str(proc_idx))
        os.environ.setdefault('WORLD_SIZE', str(args.world_size))
        rank, local, world = init_distributed(args.dist_backend if torch.cuda.is_available() else 'gloo')
        args.rank = rank
        args.batch_size = int(args.batch_size / ngpus_per_node)
        args.num_workers = int((args.num_workers + ngpus_per_node - 1) / ngpus_per_node)
    if torch.cuda.is_available():
        torch.cuda.set_device(proc_idx % torch.cuda.device_count())
    args.classes = len(trainset.classes)
    net, ema_net = init_network(args, classes=trainset.classes, old_classes=load_old_classes(args))
    # --hip_graph in a distributed run: the bare module goes to train_epoch, whose GraphedNetwork averages the gradients itself
    model = wrap_ddp(net, proc_idx) if (args.distributed and not getattr(args, 'hip_graph', False)) else net
    return train_net(model, trainset, testset, args, ema_net, fold_idx=fold_idx)


def load_old_classes(args):
    """--old_classes: yaml list of the checkpoint's classes, sorted (train_ddp.py:647-654)."""
    path = getattr(args, 'old_classes', None)
    if path is None:
        return None
    if isinstance(path, (list, tuple)):
        return sorted(path)
    import yaml
    with open(path) as f:
        args.old_classes = sorted(yaml.load(f, Loader=yaml.SafeLoader))
    return args.old_classes


def init_network(args, classes=None, old_classes=None):
    """init_network (train_ddp.py:552-590): the model is built for the OLD class list when the output layer is to be rebuilt and no separate
    pretrained file is named (the checkpoint has to load first), then `update_output_layer_onk` swaps the heads for `classes`.  The EMA copy is taken
    after the surgery (the reference rebuilds it separately; its first update, alpha = 0 at step 0, overwrites it with the parameters anyway)."""
    update = getattr(args, 'update_output_layer', False)
    if update and old_classes is None:
        raise ValueError('--update_output_layer needs --old_classes')
    from .model.utils import get_model
    src = getattr(args, 'pretrained', None)
    if update and not src and getattr(args, 'resume', False):
        update = False       # a resumed run: `latest` was saved after the surgery and already holds the new-class heads (train_net loads it)
    c = old_classes if update else classes
    net = get_model(args, pretrain=args.pretrain, classes=c)
    if update:
        # the old-class weights have to be IN the network before its heads are rebuilt (the reference's get_model(pretrain=True) loads args.pretrained before
        # update_output_layer_onk, medformer.py:224-319): --pretrained FILE is loaded here.  Without it the surgery would copy rows of a freshly initialised
        # head -- refuse that instead of doing it silently.
        if not src:
            raise ValueError('--update_output_layer rebuilds the heads of a TRAINED old-class network: name its checkpoint with --pretrained FILE')
        load_checkpoint(src, net)
        if not hasattr(net, 'aux_loss'):
            raise NotImplementedError('--update_output_layer rewires MedFormer heads (model/dim3/medformer.py:224)')
        from .model.dim3.medformer import update_output_layer_onk
        net = update_output_layer_onk(net, original_classes=old_classes, new_classes=classes, copy_pancreas=getattr(args, 'no_mask', False))
    net = net.to('cuda')
    ema_net = make_ema(net) if args.ema else None
    return net, ema_net


def load_label_names(args, root_attr='data_root', required=True):
    """Sorted class names from <root>/list/label_names.yaml (dataset_abdomenatlas_UFO.py:289-300)."""
    import yaml
    root = getattr(args, root_attr, None)
    path = os.path.join(root, 'list', 'label_names.yaml') if root else None
    if path is None or not os.path.exists(path):
        if required:
            raise ValueError('class names not found: %s/list/label_names.yaml' % root)
        return None
    with open(path) as f:
        return sorted(yaml.load(f, Loader=yaml.SafeLoader))


def main(argv=None):
    args = get_parser(argv)
    logging.basicConfig(level=logging.INFO, format='%(message)s')
    from .training.dataset import SyntheticUFODataset, AugmentedCropDataset
    if args.synthetic:
        names = [f'organ_{i}' for i in range(args.classes - 2)] + ['pancreas', 'pancreatic_lesion']
        trainset = SyntheticUFODataset(sorted(names), size=args.training_size[0], length=args.synthetic)
    elif args.load_augmented:
        trainset = AugmentedCropDataset.from_directory(args.save_destination, load_label_names(args), packed=True,
                                                       classes_ufo=load_label_names(args, 'UFO_root', required=False))
    else:
        raise SystemExit('rsuper_amd.train_ddp trains from pre-saved crops: pass --load_augmented --save_destination DIR '
                         '(or --synthetic N for a smoke run); cropping raw volumes is offline preprocessing (INTEGRATION.md)')
    return main_worker(int(os.environ.get('LOCAL_RANK', 0)), max(torch.cuda.device_count(), 1), 0, args, trainset=trainset)


if __name__ == '__main__':
    main()
