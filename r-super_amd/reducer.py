"""GradReducer -- bucketed gradient all-reduce for the one-process-per-GPU data-parallel step (train_ddp.py:623-668).

Same mathematics as torch DDP (every rank ends the backward pass with the mean of all ranks' gradients; parameters are
broadcast from rank 0 at construction), laid out for this step:
  * gradients live in a few flat f32 buckets (reverse registration order ~ the order backward produces them); the HIP
    weight-gradient kernels write straight into bucket views (`ops.grad_dest`), so there is no per-parameter copy;
  * a post-accumulate hook only counts; when a bucket is complete ONE asynchronous all-reduce is issued -- on RCCL from
    the weight-gradient side stream, so the collective is ordered after the kernels that filled the bucket and overlaps
    the data-gradient chain on the main stream (torch DDP's per-parameter hooks on the main stream rule that overlap out
    and cost ~1 ms of host/stream bookkeeping per step here);
  * `finish()` (after backward, before clip/optimiser) makes the main stream wait for the collectives.
RCCL averages in the collective (ReduceOp.AVG); gloo (CPU tests) sums and divides.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('flat', 'params', 'pending', 'handle', 'wire')

    def __init__(self, flat, params):
        self.flat, self.params, self.pending, self.handle, self.wire = flat, params, len(params), None, None


class GradReducer:
    def __init__(self, module, bucket_mb=48, process_group=None, broadcast=True, tail_mb=8, wire_dtype=None):
        """wire_dtype=torch.bfloat16 (RSUPER_DDP_BF16=1 through wrap_ddp): the buckets travel as bf16 -- one cast of the flat f32 bucket before the
        collective, one back after it; half the bytes on the xGMI ring (162 -> 81 MB per step for the base-32 UNet), gradients rounded to bf16
        before the mean (2^-9 relative per element; tests/reducer_gloo_worker.py bounds it against the f32 buckets).  Default: f32, as DDP."""
        self.wire_dtype = wire_dtype if wire_dtype not in (None, torch.float32) else None
        assert dist.is_available() and dist.is_initialized(), 'init_distributed() first'
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        params = [p for p in module.parameters() if p.requires_grad]
        assert params and all(p.dtype == torch.float32 for p in params), 'parameters are f32 in the reference layout'
        if broadcast:                                           # DDP's initial parameter broadcast (train_ddp.py:663)
            for p in params:
                dist.broadcast(p.data, 0, group=process_group)
        cap = int(bucket_mb * (1 << 20) // 4)
        groups, cur, n = [], [], 0
        for p in reversed(params):
            if cur and n + p.numel() > cap:
                groups.append(cur); cur, n = [], 0
            cur.append(p); n += p.numel()
        if cur:
            groups.append(cur)
        # The bucket that completes LAST (first-registered layers: stem / inc / down1) has nothing left to hide behind, so
        # its ring time is exposed at the end of backward: keep it small (<= tail_mb) by splitting it off the last group.
        tcap = int(tail_mb * (1 << 20) // 4)
        last = groups[-1]
        if tail_mb and sum(p.numel() for p in last) > tcap and len(last) > 1:
            k, n = len(last), 0
            while k > 1 and n + last[k - 1].numel() <= tcap:
                k -= 1; n += last[k].numel()
            if 0 < k < len(last):
                groups[-1:] = [last[:k], last[k:]]
        self.buckets, self._slot = [], {}
        for g in groups:
            flat = torch.zeros(sum(p.numel() for p in g), device=g[0].device, dtype=torch.float32)
            b = _Bucket(flat, g)
            off = 0
            for p in g:
                self._slot[p] = (b, off)
                off += p.numel()
            self.buckets.append(b)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self._dest_registered = False
        if params[0].is_cuda:
            from .hip import ops
            ops.GRAD_DEST = self.view_for                       # kernels write gradients straight into the buckets
            self._dest_registered = True

    # ------------------------------------------------------------------ bucket views
    def view_for(self, p):
        """A fresh view tensor of p's slot (a new tensor object every time, so autograd can adopt it as p.grad)."""
        slot = self._slot.get(p)
        if slot is None:
            return None
        b, off = slot
        return b.flat[off:off + p.numel()].view(p.shape)

    # ------------------------------------------------------------------ hooks
    def __deepcopy__(self, memo):
        # make_ema(net) after wrap_ddp deep-copies the module's attributes: the EMA copy must not own buckets or hooks
        return None

    def reset(self):
        """Re-arm after a failed / interrupted backward pass: waits for in-flight collectives and restores the counters."""
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
            b.pending = len(b.params)

    def _on_grad(self, p):
        b, off = self._slot[p]
        if b.pending <= 0:
            raise RuntimeError('GradReducer: a second gradient arrived for a parameter before finish() -- exactly one backward pass per '
                               'finish() is supported (no gradient accumulation; call reducer.reset() after an aborted backward)')
        g = p.grad
        if g.data_ptr() != b.flat.data_ptr() + 4 * off:         # produced elsewhere (ATen op, CPU tests): move it in
            v = self.view_for(p)
            v.copy_(g)
            p.grad = v
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _wire(self, b):
        if self.wire_dtype is None:
            return b.flat
        b.wire = b.flat.to(self.wire_dtype)                     # ordered on the launching stream, like the collective itself
        return b.wire

    def _launch(self, b):
        op = dist.ReduceOp.AVG if self.backend == 'nccl' else dist.ReduceOp.SUM
        if b.flat.is_cuda:
            from .hip import ops
            side = ops.side_stream() if ops.overlap_enabled() else None
            if side is not None:
                ev = torch.cuda.Event()
                ev.record()                                     # gradients produced on the main stream (stem, head, copies)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    b.handle = dist.all_reduce(self._wire(b), op=op, group=self.pg, async_op=True)
                return
        b.handle = dist.all_reduce(self._wire(b), op=op, group=self.pg, async_op=True)

    def finish(self):
        """Call after backward(): waits for every bucket (stream-wise on GPU) and re-arms the counters."""
        for b in self.buckets:
            if b.pending != 0:
                raise RuntimeError('a parameter received no gradient in this backward pass (find_unused_parameters=False semantics)')
            b.handle.wait()
            b.handle = None
            if self.wire_dtype is not None:
                b.flat.copy_(b.wire)                            # back into the f32 bucket the optimiser reads (after the wait: stream-ordered)
                if b.wire.is_cuda:                              # allocated on the side stream, read here on the current one: the caching
                    b.wire.record_stream(torch.cuda.current_stream())   # allocator must not recycle it for a side-stream tensor before this copy ran
                b.wire = None
            if self.backend != 'nccl':
                b.flat.div_(self.world)
            b.pending = len(b.params)

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self._dest_registered:
            from .hip import ops
            ops.GRAD_DEST = None
