"""hipGraph replay of the training step (SURVEY 8 rows a/b; the "HIP graphs instead of a tracing compiler" idiom of the MI355X port).

One iteration of train_epoch (rsuper_train/train_ddp.py:308-357: zero_grad -> net(img) -> calculate_loss -> backward -> clip ->
AdamW -> EMA) is captured ONCE into a hipGraph (torch.cuda.CUDAGraph records the raw-stream launches of the ctypes kernels like any
other stream work) and replayed for every later batch of the same shape: the host enqueues one graph launch instead of 330 (UNet) /
3000 (MedFormer) kernel launches.  What keeps a replay equal to an eager step:

  * the batch is copied into static device buffers before the replay;
  * the step-dependent scalars of the optimiser -- lr (epoch schedule), the two Adam bias corrections, the EMA alpha ramp
    (training/utils.py:135-140) -- live in a 4-float device buffer the host refreshes before each replay
    (`rsuper_adamw_ema_step_dyn`); the optimiser's Python-side step counters are advanced after the replay;
  * `ops.WEIGHTS_EPOCH` is bumped after each replay (the graph rewrites the parameters behind the caches' back).

Only steps without host-side control flow can be captured: segmentation-only supervision (`report_volume_loss_basic == 0`, the
headline configuration).  The ball search of the report losses reads device flags on the host (losses_foundation.py:1423-1492), so
batches with report supervision, DDP-wrapped modules and the sanity-check asserts take the eager `train_step`.
The first `warmup` calls run eagerly (they are real training steps: allocator pools, lazily created optimiser state and kernel
attributes are settled before capture).

`GraphedNetwork` is the form for everything else -- report supervision, the actual R-Super use: the network's forward and its backward
are captured as TWO hipGraphs (static input / output / gradient buffers, both recorded on one capture stream), the loss between them -- with the ball search's host reads -- runs eagerly on the graph's static logits, and so do
clip / AdamW / EMA.  The host then enqueues 2 graph launches + the loss + the optimiser instead of every kernel of the network
(MedFormer: ~2000 of its 2170 launches).  Under torch.distributed it also averages the gradients over the ranks after the backward replay
(`exchange_gradients`: flat buckets, one RCCL all-reduce each), so it replaces the DDP wrapper rather than sitting inside it.
"""
import math
import os

import torch
import torch.nn as nn

from .hip import ops
from .train_ddp import train_step
from .training import losses_foundation as lf
from .training.dataset.packed import PackedBits
from .training.utils import FusedAdamWEMA, ema_alpha_for_step


def _release_cached_blocks():
    """Before every capture: collect garbage and hand the caching allocator's free blocks back to the driver (a collection that frees device
    tensors INSIDE a capture aborts the process; and the capture's private pool then starts from a clean allocator).  Introduced while the replay
    corruption was still attributed to rocBLAS; the real cause were captured memset nodes.  RSUPER_CAPTURE_KEEP_CACHE=1 keeps
    the cache (A/B)."""
    if os.environ.get('RSUPER_CAPTURE_KEEP_CACHE') == '1':
        return
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _capture_safe_blas():
    """hipBLASLt (torch's default on ROCm) for the library GEMMs a capture records; MedFormer's per-call library choice off.
    History: while the "garbage after a few replays" defect was being bisected, runs with rocBLAS preferred were corrupted in 6 of 15 cases and
    runs with hipBLASLt in 0 of 13, and this switch was taken for the fix.  It was not: the victim was a captured `hipMemsetAsync` node of this
    repo's own gradient-norm accumulator, which writes 0xC0 bytes after eager interludes (removed in csrc/optim.hip) -- with that
    fixed, rocBLAS inside the capture is correct too (5 of 5 runs, RSUPER_CAPTURE_UNSAFE_BLAS=1).  The switch stays because it costs nothing
    (the pointwise products run on csrc/pointwise.hip) and keeps ADVICE r02's point: no process-wide BLAS preference survives a capture."""
    if os.environ.get('RSUPER_CAPTURE_UNSAFE_BLAS') == '1':      # A/B of the defect: leave whatever library is preferred
        return
    if torch.cuda.is_available() and hasattr(torch.backends.cuda, 'preferred_blas_library'):
        torch.backends.cuda.preferred_blas_library('cublaslt')
    try:
        from .model.dim3 import medformer_utils as _mu
        _mu.gemm_library.active = False
    except Exception:
        pass


class GraphedTrainStep:
    def __init__(self, net, ema_net, optimizer, args, classes, warmup=3):
        if not isinstance(optimizer, FusedAdamWEMA) or len(optimizer.param_groups) != 1:
            raise ValueError('GraphedTrainStep needs the fused AdamW+EMA optimiser with one parameter group')
        if getattr(net, '_rsuper_reducer', None) is not None or hasattr(net, 'module'):
            raise ValueError('data-parallel modules take the eager step (the gradient exchange is not captured)')
        if float(getattr(args, 'report_volume_loss_basic', 0.0)) > 0:
            raise ValueError('report supervision has host-side control flow (ball search): use the eager train_step')
        if int(warmup) < 1:
            # a fresh optimiser creates exp_avg / exp_avg_sq (zeros_like) and its norm scratch lazily in the first step: inside the capture
            # those zero-fills would be recorded and both moments reset on every replay
            raise ValueError('GraphedTrainStep needs warmup >= 1: the optimiser state must exist before the capture')
        self.net, self.ema, self.opt, self.args, self.classes = net, ema_net, optimizer, args, list(classes)
        self.warmup, self.calls = int(warmup), 0
        self.graph, self.static, self.out = None, None, None
        self.dyn = None
        self._replays = 0
        on = os.environ.get('RSUPER_GRAPH_VERIFY', '1') != '0'
        self.verify_at, self.verify_every = ((1, 12, 50) if on else ()), (1000 if on else 0)

    # ------------------------------------------------------------------------------------------------------------------
    def _scalars(self, step):
        g = self.opt.param_groups[0]
        # the eager entry point receives lr / betas as C floats and forms the bias corrections in double from those (csrc/api.hip
        # rsuper_adamw_ema_step): the same roundings here keep a replay bit-identical to an eager step
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
        lr, b1, b2 = f32(g['lr']), f32(g['betas'][0]), f32(g['betas'][1])
        t = self._opt_step() + 1
        ema_on = self.ema is not None and getattr(self.args, 'ema', True)
        return [lr, lr / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t),
                float(ema_alpha_for_step(getattr(self.args, 'ema_alpha', 0.99), step)) if ema_on else 0.0]

    def _opt_step(self):
        for p in self.opt.param_groups[0]['params']:
            st = self.opt.state.get(p)
            if st and 'step' in st:
                return int(st['step'])
        return 0

    def _set_opt_step(self, t):
        for p in self.opt.param_groups[0]['params']:
            st = self.opt.state.get(p)
            if st is not None and 'step' in st:
                st['step'] = t

    # ------------------------------------------------------------------------------------------------------------------
    def __call__(self, batch, step):
        """Same contract as train_step(net, ema, opt, batch, args, classes, step): returns (loss dict, pre-clip gradient norm).  The
        returned tensors are static buffers of the graph -- read them before the next call."""
        self.calls += 1
        if self.calls <= self.warmup:
            # detached: a loss tensor kept by the caller would keep this step's autograd graph -- and its AccumulateGrad nodes, which are
            # bound to the default stream -- alive into the capture, where the engine would then synchronise the default stream with the
            # capturing one (hipStreamEndCapture crashes on that join)
            loss, gnorm = train_step(self.net, self.ema, self.opt, batch, self.args, self.classes, step)
            return {k: v.detach() for k, v in loss.items()}, gnorm
        if self.graph is None:
            self._capture(batch, step)
        else:
            for k, v in batch.items():
                dst = self.static[k]
                if isinstance(dst, PackedBits) != isinstance(v, PackedBits):
                    raise ValueError(f'batch entry {k!r} changed form (bit-packed vs uint8) after the capture')
                if isinstance(v, PackedBits):        # the bit-packed label of segmentation-only batches (dataset/packed.py): the graph reads dst.packed
                    if dst.C != v.C:
                        raise ValueError(f'batch entry {k!r} changed its class count: {v.C} vs captured {dst.C}')
                    dst.reset()                      # planes / flags inflated outside the graph (self-verification) belong to the previous batch
                    dst, v = dst.packed, v.packed
                if dst.shape != v.shape or dst.dtype != v.dtype:
                    raise ValueError(f'batch entry {k!r} changed shape / dtype: {tuple(v.shape)} {v.dtype} vs captured {tuple(dst.shape)} {dst.dtype}')
                dst.copy_(v, non_blocking=True)
        # a FRESH pinned staging tensor per step: with one reused pinned buffer a host that runs ahead (no per-step synchronisation) overwrote
        # the scalars of step i with those of step i + 1 before the asynchronous copy of step i had executed -- the replay then used the next
        # step's bias corrections (losses drifted from the eager run after a few dozen steps).  The pinned allocator only recycles a block
        # once the copies recorded on it have completed.
        self.dyn.copy_(torch.tensor(self._scalars(step), dtype=torch.float32).pin_memory(), non_blocking=True)
        t = self._opt_step() + 1
        self._replays += 1
        check = self._replays in self.verify_at or (self.verify_every and self._replays % self.verify_every == 0)
        before = [p.detach().clone() for p in self._params()] if check else None
        self.graph.replay()
        self._set_opt_step(t)
        ops.WEIGHTS_EPOCH += 1
        if check:
            self._verify(before)
        if os.environ.get('RSUPER_GRAPH_DEBUG') == '1':          # which gradient buffers of this replay are not finite (debugging aid, synchronises)
            names = {id(p): k for k, p in self.net.named_parameters()}
            bad = [(names.get(id(p), '?'), float(p.grad.abs().max())) for p in self._params() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            print(f'[graph debug] replay {self._replays}: {len(bad)} non-finite gradients {bad[:6]}', flush=True)
        return self.out

    def _params(self):
        return [p for p in self.opt.param_groups[0]['params'] if p.requires_grad]

    def _verify(self, before):
        """The replayed step's gradients (still in the graph's gradient buffers) against an eager forward + loss + backward from the same
        batch and the PRE-step parameters -- the check GraphedNetwork runs, for the whole-step graph: a captured ATen reduction returned
        garbage from the 12th replay on in round 2, silently.  Replays 1, 12, 50 and every 1000th; costs one parameter
        copy each way and one eager forward + backward, whose activations live outside the graph pool (about one more step's worth of
        memory at those replays; on OutOfMemoryError the check is skipped with a warning and the updated parameters are kept);
        RSUPER_GRAPH_VERIFY=0 turns it off."""
        params = self._params()
        grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
        after = [p.detach().clone() for p in params]
        sanity = lf.SANITY_CHECKS
        lf.SANITY_CHECKS = False
        try:
            with torch.no_grad():
                torch._foreach_copy_(params, before)
            ops.WEIGHTS_EPOCH += 1
            with torch.enable_grad():
                b = self.static
                result = self.net(b['image'])
                loss_all = lf.calculate_loss(model_output=result, label=b['label'], unk_voxels=b.get('unk_channels'), args=self.args, matcher=None,
                                             chosen_segment_mask=b.get('mask'), tumor_volumes_report=b.get('volumes'),
                                             tumor_diameters=b.get('diameters'), classes=self.classes, input_tensor=b['image'],
                                             class_weights=b.get('weights'))
                ref = torch.autograd.grad(loss_all['overall'], params, allow_unused=True)
        except torch.cuda.OutOfMemoryError:
            # the eager pass allocates a second set of activations OUTSIDE the graph's private pool (which still holds the replayed step's): a
            # run sized to fit the replayed step only cannot afford it.  Skip this check rather than end the training run hours in.
            import warnings
            warnings.warn(f'GraphedTrainStep: not enough memory for the self-verification of replay {self._replays} (an eager forward + backward '
                          'next to the graph pool); check skipped.  RSUPER_GRAPH_VERIFY=0 turns the checks off.')
            ref = None
        finally:
            lf.SANITY_CHECKS = sanity
            with torch.no_grad():
                torch._foreach_copy_(params, after)
            ops.WEIGHTS_EPOCH += 1
            del after, before
            result = loss_all = None
        if ref is None:
            torch.cuda.empty_cache()
            return
        names = {id(p): k for k, p in self.net.named_parameters()}
        for p, g, r in zip(params, grads, ref):
            if g is None or r is None:
                continue
            scale = float(r.abs().max())
            err = float((g - r).abs().max())
            if not (err <= 1e-3 * max(scale, 1e-30) or err <= 1e-12):
                raise RuntimeError(f'GraphedTrainStep: replay {self._replays} disagrees with the eager step on the gradient of '
                                   f'{names.get(id(p), "?")} (max difference {err:.3e}, largest entry {scale:.3e}); the parameters have already '
                                   'been updated with it -- reload the last checkpoint and run without --hip_graph '
                                   '(tools/mode_consistency.py narrows the kernel down)')

    def _capture(self, batch, step):
        dev = next(self.net.parameters()).device
        self.static = {k: (PackedBits(v.packed.to(dev).clone(), v.C) if isinstance(v, PackedBits) else v.to(dev).clone()) for k, v in batch.items()}
        self.dyn = torch.zeros(4, device=dev, dtype=torch.float32)
        self.dyn.copy_(torch.tensor(self._scalars(step), dtype=torch.float32))
        sanity = lf.SANITY_CHECKS
        lf.SANITY_CHECKS = False                     # its asserts read device flags on the host: not capturable (train_epoch keeps its own)
        t0 = self._opt_step()
        self.opt.dyn = self.dyn
        try:
            torch.cuda.synchronize()
            _release_cached_blocks()
            _capture_safe_blas()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                loss, gnorm = train_step(self.net, self.ema, self.opt, self.static, self.args, self.classes, step)
            self.out = ({k: v.detach() for k, v in loss.items()}, gnorm)
        finally:
            lf.SANITY_CHECKS = sanity
            self.opt.dyn = None
            self._set_opt_step(t0)                   # capture only records: the counters advance when the graph is replayed


def exchange_gradients(grads, group=None, bucket_bytes=64 << 20, force=False):
    """Mean of `grads` (list of tensors, modified in place) over the ranks of `group`: the gradient exchange of DistributedDataParallel
    (train_ddp.py:663) for gradients that already sit in their final buffers -- flatten a bucket, ONE all-reduce per bucket (launched
    asynchronously, all buckets in flight together), copy back.  RCCL averages in the collective; gloo (CPU tests) sums and divides."""
    import torch.distributed as dist
    from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
    world = dist.get_world_size(group)
    if (world == 1 and not force) or not grads:           # force: run the collectives on a one-rank group too (tests)
        return
    avg = dist.get_backend(group) == 'nccl'
    buckets, cur, n = [], [], 0
    for g in grads:
        if cur and n + g.numel() * g.element_size() > bucket_bytes:
            buckets.append(cur); cur, n = [], 0
        cur.append(g); n += g.numel() * g.element_size()
    if cur:
        buckets.append(cur)
    work = []
    for b in buckets:
        flat = _flatten_dense_tensors(b)
        work.append((b, flat, dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True)))
    for b, flat, h in work:
        h.wait()
        if not avg:
            flat.div_(world)
        torch._foreach_copy_(b, list(_unflatten_dense_tensors(flat, b)))


class _TupleOut(nn.Module):
    """The network with a tuple of tensors as output (what make_graphed_callables can hold in static buffers)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, img):
        seg = self.net(img)['segmentation']
        return tuple(seg) if isinstance(seg, (list, tuple)) else (seg,)


class _ReplayFn(torch.autograd.Function):
    """Forward: copy the input into the static buffer, replay the forward graph, hand out the static outputs.  Backward: copy the output
    gradients into their static buffers, replay the backward graph and point every parameter's .grad at its static gradient buffer.
    The parameters are not inputs of this node on purpose: routed through AccumulateGrad their gradients would be cloned one by one
    (the static buffers are referenced twice, so the engine cannot steal them) -- 250 extra launches per MedFormer step."""

    @staticmethod
    def forward(ctx, anchor, owner, img):
        owner.static_img.copy_(img)
        owner.fwd_graph.replay()
        ctx.owner = owner
        return tuple(o.detach() for o in owner.static_outs)

    @staticmethod
    def backward(ctx, *gouts):
        o = ctx.owner
        probe = next((i for i, g in enumerate(o.static_grads) if g is not None), None)
        if probe is not None and o.params[probe].grad is o.static_grads[probe] and o.static_grads[probe]._version == o._grad_version:
            # .grad still points at the static buffer and nothing reset it since the last backward: a second backward before the optimiser
            # step (micro-batch accumulation) would be overwritten by this replay
            raise RuntimeError('GraphedNetwork: one backward per zero_grad / optimiser step (gradient accumulation needs the eager module)')
        for dst, g in zip(o.static_gouts, gouts):
            dst.copy_(g) if g is not None else dst.zero_()
        o.bwd_graph.replay()
        o._replays += 1
        if o._replays in o.verify_at or (o.verify_every and o._replays % o.verify_every == 0):
            o._verify()
        if o.exchange:
            exchange_gradients([g for g in o.static_grads if g is not None], o.group, force=True)
        for p, g in zip(o.params, o.static_grads):
            if g is None:
                continue
            if p.grad is None or p.grad is g:       # `is g`: zero_grad(set_to_none=False) zeroed the static buffer, the replay refilled it
                p.grad = g
            else:
                p.grad.add_(g)                      # gradient accumulation into a tensor of the caller's
        o._grad_version = o.static_grads[probe]._version if probe is not None else -1
        return None, None, None


class GraphedNetwork:
    """`net(img)` -> {'segmentation': logits | [logits, aux]} with the forward and the backward of the network replayed from two hipGraphs.
    Drop-in for the module in `train_step(net, ...)` for one input shape; parameters, buffers and `state_dict` stay those of `net`
    (attribute access falls through).  The first training call captures, after `warmup` eager forward / backward iterations on the
    capture stream (real kernels, but no optimiser step: parameters are untouched).  Gradients land in static buffers that the
    parameters' .grad point at after every backward (`zero_grad(set_to_none=True)` only drops the reference)."""

    def __init__(self, net, warmup=3, exchange=None, process_group=None):
        """exchange: average the gradients over the ranks of `process_group` after every backward replay (None: whenever torch.distributed
        is initialised with more than one rank).  The module must be the bare network -- not wrapped by DistributedDataParallel or
        `wrap_ddp`: the exchange happens here, on the static gradient buffers, after the backward graph (no overlap with backward, but the
        host no longer enqueues the network's launches one by one); parameters are broadcast from rank 0 like DDP does at construction."""
        if getattr(net, '_rsuper_reducer', None) is not None or hasattr(net, 'module'):
            raise ValueError('GraphedNetwork takes the bare module (it exchanges the gradients itself when torch.distributed is initialised)')
        import torch.distributed as dist
        self.net, self.warmup = net, int(warmup)
        self.group = process_group
        if exchange is None:
            exchange = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.exchange = bool(exchange)
        if self.exchange:
            with torch.no_grad():
                for t in list(net.parameters()) + list(net.buffers()):
                    dist.broadcast(t, 0, group=process_group)
            ops.WEIGHTS_EPOCH += 1
        self.fwd_graph = None
        self._shape = None
        self._anchor = None
        self._replays = 0
        on = os.environ.get('RSUPER_GRAPH_VERIFY', '1') != '0'
        self.verify_at, self.verify_every = ((1, 12, 50) if on else ()), (1000 if on else 0)

    def __getattr__(self, name):                       # parameters(), named_parameters(), state_dict(), train(), eval(), ...
        return getattr(self.__dict__['net'], name)

    def _verify(self):
        """Recompute this step's gradients eagerly (same input buffer, same output gradients) and compare them with what the backward graph
        left in the static buffers.  The step is bit-reproducible across eager and replayed execution, so anything beyond summation-order
        noise is a defect of the replay -- like the captured ATen reduction that returned garbage from the 12th replay on,
        which this check reports at replay 12 instead of letting one bias train on noise.  Runs at replays 1, 12, 50 and every 1000th
        (one extra eager forward + backward each); RSUPER_GRAPH_VERIFY=0 turns it off."""
        with torch.enable_grad():
            outs = self._wrapped(self.static_img)
            ref = torch.autograd.grad(outs, self.params, self.static_gouts, allow_unused=True)
        names = {id(p): k for k, p in self.net.named_parameters()}
        for p, g, r in zip(self.params, self.static_grads, ref):
            if g is None or r is None:
                continue
            scale = float(r.abs().max())
            err = float((g - r).abs().max())
            if err > 1e-3 * max(scale, 1e-30) and err > 1e-12:
                raise RuntimeError(f'GraphedNetwork: replay {self._replays} disagrees with the eager step on the gradient of '
                                   f'{names.get(id(p), "?")} (max difference {err:.3e}, largest entry {scale:.3e}); '
                                   'run without --hip_graph and report the kernel (tools/mode_consistency.py narrows it down)')
        ops.WEIGHTS_EPOCH += 1

    def _capture(self, img):
        _capture_safe_blas()                           # before the warm-up iterations too: they choose the GEMM algorithms the capture records
        wrapped = self._wrapped = _TupleOut(self.net)
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        saved = [p.grad for p in self.params]
        self.static_img = img.detach().clone()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(self.warmup):
                outs = wrapped(self.static_img)
                torch.autograd.backward(outs, [torch.zeros_like(o) for o in outs])
                for p in self.params:
                    p.grad = None
                del outs
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        _release_cached_blocks()
        _capture_safe_blas()
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.fwd_graph, pool=pool, stream=stream):
            outs = wrapped(self.static_img)
        self.static_outs = tuple(outs)
        self.static_gouts = [torch.zeros_like(o) for o in outs]
        with torch.cuda.graph(self.bwd_graph, pool=pool, stream=stream):
            torch.autograd.backward(self.static_outs, self.static_gouts)
        self.static_grads = [p.grad for p in self.params]
        for p, g in zip(self.params, saved):
            p.grad = g
        self._anchor = torch.zeros(1, device=img.device, requires_grad=True)
        self._grad_version = -1
        self._shape = (tuple(img.shape), img.dtype)
        ops.WEIGHTS_EPOCH += 1                         # fragment caches filled during capture belong to the graphs' replays

    def accepts(self, img):
        """Not captured yet, or captured for this input shape / dtype."""
        return self._shape is None or (tuple(img.shape), img.dtype) == self._shape

    def __call__(self, img):
        if not self.net.training or not torch.is_grad_enabled():
            return self.net(img)
        if self.fwd_graph is None:
            self._capture(img)
        if (tuple(img.shape), img.dtype) != self._shape:
            raise ValueError(f'GraphedNetwork was captured for input {self._shape}, got {(tuple(img.shape), img.dtype)}'
                             + (' (with the gradient exchange on there is no eager fall-back: keep the batch shape fixed, drop_last)' if self.exchange else ''))
        out = _ReplayFn.apply(self._anchor, self, img)
        return {'segmentation': list(out) if len(out) > 1 else out[0]}
