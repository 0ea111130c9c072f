"""Autograd functions over the C ABI (include/rsuper_hip.h).  PyTorch supplies device memory, the current HIP
stream and the autograd graph; every device op below is a hand-written gfx950 kernel in csrc/.

Activation convention between functions: a contiguous tensor of shape (N, D, H, W, C) (channels-last storage,
bf16 or f32) plus its InstanceNorm statistics `mr` of shape (N, C, 2) = (mean, rstd), f32, computed by the
producing kernel's epilogue (model/dim3/conv_layers.py:40-42: eps=1e-4, affine=False, biased variance).
"""
import os

import torch
import torch.nn.functional as F

from . import lib as _l

EPS = 1e-4
_DT = {torch.float32: _l.F32, torch.bfloat16: _l.BF16}


def use_tr():
    """bf16 weight-gradient operand fetch via ds_read_b64_tr_b16 (1, default) or 16-bit LDS reads (0)."""
    return int(os.environ.get('RSUPER_WGRAD_TR', '1'))


_DEV_INDEX = None


def _stream():
    """Raw handle of the calling thread's current HIP stream.  torch.cuda.current_stream() builds a Python Stream object per
    call (~8 us, ~110 launches per forward): the C accessor returns the same handle in well under a microsecond."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()           # one device per process (set before the first launch)
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def _ptr(t, off_elems=0):
    if t is None:
        return None
    return t.data_ptr() + off_elems * t.element_size()


_WS = None


def _L():
    """The C library, with the split-reduction workspace of the 6^3-level convolutions registered on first use on a GPU.  One device per
    process (one rank per GPU): the workspace lives on the device the first launch pins (`_stream`), the same one every later launch uses."""
    global _WS, _DEV_INDEX
    L = _l.lib()
    if _WS is None and torch.cuda.is_available():
        if _DEV_INDEX is None:
            _DEV_INDEX = torch.cuda.current_device()
        _WS = torch.empty((L.rsuper_conv3_workspace_bytes(),), device=torch.device('cuda', _DEV_INDEX), dtype=torch.uint8)
        _l.check(L.rsuper_conv3_set_workspace(_WS.data_ptr(), _WS.numel()), 'conv3_set_workspace')
    return L


def pick_bn(n_cols, dtype, tiles_total=None, dims=None, epi=None, mixed=False):
    """Block N tile of the implicit GEMM (32/64/128 output columns); f32 parity mode is limited to 64.
    tiles_total (spatial tiles x batch): on small volumes prefer the largest tile that still yields >= 512 workgroups
    (24^3 and below would otherwise leave most of the 256 CUs idle; the narrower tiles also run on the persistent kernel).
    dims = (N, D, H, W): launches the library would hand to the volume-fitted K-split kernel (rsuper_conv3_box_bn: low-resolution
    levels that cannot fill the chip with 4x4x16 tiles) take its 64-column blocks.
    epi (0 forward, 1 data gradient; with dims): launches the library hands to the depth-reuse kernel (rsuper_conv3_kd_bn: the wide full-resolution
    layers) take its 64 / 96 / 128-column blocks.  mixed: the launch has one normalised and one raw source (the depth-reuse kernel does not take those;
    pass the same flag to part_buffer)."""
    cands = (32, 64) if dtype == torch.float32 else (32, 64, 128)
    if dims is not None and epi is not None and dtype != torch.float32:
        bn = _L().rsuper_conv3_kd_bn(_DT[dtype], epi, *dims, n_cols, 1 if mixed else 0)
        if bn:
            return bn
    if dtype != torch.float32 and dims is None and _L().rsuper_conv3_variant(-1) in (6, 7):
        return 64         # forced volume-fitted kernel (tests): 64-column blocks whatever the column count
    if dims is not None and dtype != torch.float32:
        bn = _L().rsuper_conv3_box_bn(_DT[dtype], *dims, n_cols)
        if bn:
            return bn
    if n_cols <= 32:
        return 32
    if dtype != torch.float32 and _L().rsuper_conv3_variant(-1) == 4:
        return 32         # forced weight-stationary kernel (tests / experiments): 32-column blocks, Cout / 32 of them in grid.y
    fill = int(os.environ.get('RSUPER_BN_FILL', '512'))       # two resident blocks per CU (measured 128 / 256 / 512: 13.89 / - / 13.80 ms per step)
    if tiles_total is not None and tiles_total * -(-n_cols // 128) < fill:
        for bn in reversed(cands):
            if tiles_total * -(-n_cols // bn) >= fill:
                return bn
        return 32
    if dtype != torch.float32 and n_cols > 64 and n_cols % 64:
        return 32         # e.g. 96 columns (up4.0 data gradient): three exact 32-column tiles on the producer/consumer kernel
                          # (671 us) beat one 25 %-padded 128 tile on the classic kernel (739 us)        # e.g. 96 columns: one padded 128 tile beats three 32-column tap-split tiles
    best = None
    for bn in cands:
        padded = -(-n_cols // bn) * bn
        key = (padded, -bn)
        if best is None or key < best[0]:
            best = (key, bn)
    return best[1]


def _chk_act(x):
    assert x.dim() == 5 and x.is_contiguous() and x.dtype in _DT and x.shape[-1] % 8 == 0, (x.shape, x.dtype, x.is_contiguous())


def part_buffer(dtype, dims, n_cols, bn, device, fill=None, epi=0, mixed=False):
    """Per-block partial-sum buffer (N, rows, n_cols, 2) in the shape rsuper_conv3_igemm(epi, ...) writes for this dtype/bn."""
    N, D, H, W = dims
    rows = _L().rsuper_conv3_part_rows(_DT[dtype], epi, N, D, H, W, n_cols, bn, 1 if mixed else 0)
    if fill is None:
        return torch.empty((N, rows, n_cols, 2), device=device, dtype=torch.float32)
    return torch.full((N, rows, n_cols, 2), fill, device=device, dtype=torch.float32)


def stats_finalize(part, cnt, mode=0, split=0):
    """part (N, nblk, C, 2) f32 -> (N, C, 2); with 0 < split < C two contiguous tables (N, split, 2), (N, C - split, 2)."""
    N, nblk, C, _ = part.shape
    out = torch.empty((N, C, 2), device=part.device, dtype=torch.float32)
    _l.check(_L().rsuper_stats_finalize(_ptr(part), N, nblk, C, float(cnt), EPS, mode, split, _ptr(out), _stream()), 'stats_finalize')
    if split:
        flat = out.view(-1)
        return flat[:N * split * 2].view(N, split, 2), flat[N * split * 2:].view(N, C - split, 2)
    return out


def pack_weights(dtype, mode, wa, wb, ka, kb, na, nb, bn):
    dt = _DT[dtype]
    n = _L().rsuper_conv3_packed_elems(dt, ka, kb, na + nb, bn)
    out = torch.empty((n,), device=wa.device, dtype=dtype)
    _l.check(_L().rsuper_conv3_pack_weights(dt, mode, _ptr(wa), _ptr(wb), ka, kb, na, nb, bn, _ptr(out), _stream()), 'pack_weights')
    return out


def _spec_cols(sp):
    """GEMM-N columns of a packing spec (mode, wa, wb, ka, kb, na, nb, bn): mode 1 with nb != 0 is the column sub-range nb = first << 16 | columns."""
    return (sp[6] & 0xFFFF) if (sp[0] == 1 and sp[6]) else sp[5] + sp[6]


def pack_weights_batch(dtype, specs):
    """One launch for many layers.  specs: list of (mode, wa, wb, ka, kb, na, nb, bn).  Returns a list of packed views."""
    import ctypes
    dt = _DT[dtype]
    n = len(specs)
    sizes = [_L().rsuper_conv3_packed_elems(dt, sp[3], sp[4], _spec_cols(sp), sp[7]) for sp in specs]
    offs = [0]
    for z in sizes:
        offs.append(offs[-1] + z)
    buf = torch.empty((offs[-1],), device=specs[0][1].device, dtype=dtype)
    desc = (ctypes.c_int * (6 * n))(*[v for sp in specs for v in (sp[0], sp[3], sp[4], sp[5], sp[6], sp[7])])
    wa = (ctypes.c_void_p * n)(*[sp[1].data_ptr() for sp in specs])
    wb = (ctypes.c_void_p * n)(*[(sp[2].data_ptr() if sp[2] is not None else None) for sp in specs])
    oe = (ctypes.c_size_t * n)(*offs[:-1])
    _l.check(_L().rsuper_conv3_pack_weights_batch(dt, n, desc, wa, wb, oe, _ptr(buf), _stream()), 'pack_weights_batch')
    return [buf[offs[i]:offs[i + 1]] for i in range(n)]


_BLOCK_PACK_CACHE = {}


def block_packs(w1, w2, ws, Ca, Cb, dtype, tiles_total, with_backward, dims):
    """block_pack_specs + pack_weights_batch for one BasicBlock, with the launch descriptor (tile sizes, offsets, ctypes argument arrays) cached per
    (parameter addresses, geometry): the host builds it once instead of every step -- at the 12^3 / 6^3 levels a block's kernels take 60-70 us and
    its host code took 100 us (tools/host_profile.py), the queue ran dry there.  Parameters keep their addresses for the life of a module; a new
    key (another net, a moved parameter) simply adds an entry.  Returns (buffers, bns)."""
    import ctypes
    key = (w1.data_ptr(), w2.data_ptr(), 0 if ws is None else ws.data_ptr(), w1.shape[0], Ca, Cb, dtype, tiles_total, with_backward, dims, rs_variant_epoch())
    ent = _BLOCK_PACK_CACHE.get(key)
    if ent is None:
        specs, bns = block_pack_specs(w1, w2, ws, Ca, Cb, dtype, tiles_total, with_backward, dims)
        dt = _DT[dtype]
        n = len(specs)
        sizes = [_L().rsuper_conv3_packed_elems(dt, sp[3], sp[4], _spec_cols(sp), sp[7]) for sp in specs]
        offs = [0]
        for z in sizes:
            offs.append(offs[-1] + z)
        desc = (ctypes.c_int * (6 * n))(*[v for sp in specs for v in (sp[0], sp[3], sp[4], sp[5], sp[6], sp[7])])
        wa = (ctypes.c_void_p * n)(*[sp[1].data_ptr() for sp in specs])
        wb = (ctypes.c_void_p * n)(*[(sp[2].data_ptr() if sp[2] is not None else None) for sp in specs])
        oe = (ctypes.c_size_t * n)(*offs[:-1])
        if len(_BLOCK_PACK_CACHE) > 1024:
            _BLOCK_PACK_CACHE.clear()
        ent = _BLOCK_PACK_CACHE[key] = (dt, n, desc, wa, wb, oe, offs, bns)
    dt, n, desc, wa, wb, oe, offs, bns = ent
    buf = torch.empty((offs[-1],), device=w1.device, dtype=dtype)
    _l.check(_L().rsuper_conv3_pack_weights_batch(dt, n, desc, wa, wb, oe, _ptr(buf), _stream()), 'pack_weights_batch')
    return [buf[offs[i]:offs[i + 1]] for i in range(n)], bns


def rs_variant_epoch():
    """The igemm variant / tile-fill switches change pick_bn's answers: part of the cache key."""
    return (_L().rsuper_conv3_variant(-1), os.environ.get('RSUPER_BN_FILL', '512'), os.environ.get('RSUPER_SPLIT_DGRAD', '1'))


def block_pack_specs(w1, w2, ws, Ca, Cb, dtype, tiles_total, with_backward, dims=None):
    """The (up to) four fragment buffers a BasicBlock needs: forward conv1(+shortcut), forward conv2, data-gradient
    conv2, data-gradient conv1(+shortcut).  Returns (specs, bns) in that order."""
    Cout, Cin = w1.shape[0], Ca + Cb
    has_sc = ws is not None
    nc1 = Cout * (2 if has_sc else 1)
    bn1, bn2 = pick_bn(nc1, dtype, tiles_total, dims, epi=0), pick_bn(Cout, dtype, tiles_total, dims, epi=0)
    specs = [(0, w1, ws, Ca, Cb, Cout, Cout if has_sc else 0, bn1), (0, w2, None, Cout, 0, Cout, 0, bn2)]
    bns = [bn1, bn2]
    if with_backward:
        bnd2, bnd1 = pick_bn(Cout, dtype, tiles_total, dims, epi=1), pick_bn(Cin, dtype, tiles_total, dims, epi=1)
        specs += [(1, w2, None, Cout, 0, Cout, 0, bnd2)]
        bns += [bnd2]
        if split_dgrad_sources(Ca, Cb, dtype, tiles_total, bnd1):
            # one data-gradient launch per forward source (column ranges [0, Ca) and [Ca, Ca + Cb) of the same GEMM): entries 3 and 4
            bna, bnb = pick_bn(Ca, dtype, tiles_total, dims, epi=1), pick_bn(Cb, dtype, tiles_total, dims, epi=1)
            specs += [(1, w1, ws, Cout, Cout if has_sc else 0, Cin, Ca, bna), (1, w1, ws, Cout, Cout if has_sc else 0, Cin, (Ca << 16) | Cb, bnb)]
            bns += [bna, bnb]
        else:
            specs += [(1, w1, ws, Cout, Cout if has_sc else 0, Cin, 0, bnd1)]
            bns += [bnd1]
    return specs, bns


def split_dgrad_sources(Ca, Cb, dtype, tiles_total, bn_joint):
    """The data gradient of a two-source block (up_block: [skip | up-sampled]) as TWO launches, one per source, when the joint column count only fits
    32-column blocks (96 = 32 + 64 at up4.0: three 32-column blocks re-stage dY three times, 735 us; a 32-column and a 64-column launch 224 + 424 us on the
    same box, profiles/r05_split_dgrad.txt).  RSUPER_SPLIT_DGRAD=0 keeps the single launch."""
    if os.environ.get('RSUPER_SPLIT_DGRAD', '1') == '0' or dtype == torch.float32 or not Cb:
        return False
    return bn_joint == 32 and Ca % 32 == 0 and Cb % 64 == 0 and tiles_total is not None and tiles_total >= 2048


class Src:
    """A channels-last source view: tensor (N,D,H,W,ld) + channel offset/count + optional (N,C,2) stats."""

    def __init__(self, t, C=None, off=0, mr=None):
        self.t, self.off, self.mr = t, off, mr
        self.ld = t.shape[-1]
        self.C = self.ld - off if C is None else C

    def args(self):
        return (_ptr(self.t, self.off), self.ld, self.C, _ptr(self.mr))


_NONE = (None, 0, 0, None)


class KernelTimer:
    """Optional HIP-event timing of the MFMA conv launches on the launch stream (used by bench.py's roofline leg).
    Records (kind, algorithmic FLOPs, start event, end event) per launch; `summary()` synchronises.
    Weight gradients run on a side stream concurrently with the data-gradient chain, so per-launch durations overlap in
    wall time: `busy_ms` is the length of the UNION of all launch intervals on the device timeline (time during which at
    least one timed kernel was running) -- the denominator of an honest aggregate FLOP rate."""

    def __init__(self, fenced=None):
        # fenced=False (default): events created with hipEventDisableSystemFence through the C ABI (rsuper_timer_event_*): a default event -- what
        # torch.cuda.Event(enable_timing=True) creates -- performs a system-scope release (cache write-back + invalidate) every time it is recorded, i.e. twice per
        # bracketed launch: that is charged to the bracket and makes every timed kernel start on a cold L2 (the packed weight fragments and the halo rows its
        # predecessor left there).  hip_runtime_api.h recommends the flag for events that only measure time.  RSUPER_TIMER_FENCED=1 / fenced=True: torch events (A/B).
        self.fenced = (os.environ.get('RSUPER_TIMER_FENCED', '0') == '1') if fenced is None else bool(fenced)
        self.records = []
        self._pool = []
        self.step = 0
        self.base = self._event()
        self._record(self.base)

    def next_step(self):
        """Launches recorded from here on belong to the next step (summary() then also reports per-step figures)."""
        self.step += 1

    def _event(self):
        if self.fenced:
            return torch.cuda.Event(enable_timing=True)
        import ctypes
        h = ctypes.c_void_p()
        _l.check(_L().rsuper_timer_event_create(ctypes.byref(h)), 'timer_event_create')
        self._pool.append(h)
        return h

    def _record(self, ev):
        if self.fenced:
            ev.record()
        else:
            _l.check(_L().rsuper_timer_event_record(ev, _stream()), 'timer_event_record')

    def _elapsed(self, a, b):
        if self.fenced:
            return a.elapsed_time(b)
        import ctypes
        ms = ctypes.c_float()
        _l.check(_L().rsuper_timer_event_elapsed_ms(a, b, ctypes.byref(ms)), 'timer_event_elapsed')
        return float(ms.value)

    def launch(self, kind, flops, fn):
        s, e = self._event(), self._event()
        self._record(s)
        fn()
        self._record(e)
        self.records.append((kind, flops, s, e, self.step))

    @staticmethod
    def _union(spans):
        spans = sorted(spans)
        busy, cur_s, cur_e = 0.0, None, None
        for a, b in spans:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        if cur_e is not None:
            busy += cur_e - cur_s
        return busy

    def summary(self, tol=0.05):
        """Per-kind totals over the steps whose conv time (union of the launch intervals) lies within `tol` of the median step.  A bracket opens when its start
        event executes and closes after the kernel: whenever the host falls behind the device (a 20 ms hiccup between recording the event and launching the
        kernel was seen about once in 40 passes; the device then idles INSIDE the bracket, and the next few steps run with a drained queue) the bracket measures
        the host, not the kernel.  `step_busy_ms` lists every step, `steps_used` the ones that count."""
        torch.cuda.synchronize()
        spans, per_step, recs = [], {}, []
        for kind, flops, s, e, step in self.records:
            el = self._elapsed(s, e)
            sp = (self._elapsed(self.base, s), self._elapsed(self.base, e))
            spans.append(sp)
            per_step.setdefault(step, []).append(sp)
            recs.append((kind, flops, el, step))
        self.busy_ms_all = self._union(spans)
        steps = sorted(per_step)
        self.step_busy_ms = [self._union(per_step[k]) for k in steps]     # union of the launch intervals of every step
        sb = sorted(self.step_busy_ms)
        med = (sb[len(sb) // 2] if len(sb) % 2 else 0.5 * (sb[len(sb) // 2 - 1] + sb[len(sb) // 2])) if sb else 0.0
        good = {k for k, v in zip(steps, self.step_busy_ms) if abs(v - med) <= tol * med} if len(steps) > 1 else set(steps)
        self.steps_used = len(good)
        self.busy_ms = sum(v for k, v in zip(steps, self.step_busy_ms) if k in good)
        out = {}
        for kind, flops, el, step in recs:
            if step not in good:
                continue
            d = out.setdefault(kind, dict(launches=0, flops=0.0, ms=0.0, max_ms=0.0))
            d['launches'] += 1
            d['flops'] += flops
            d['ms'] += el
            d['max_ms'] = max(d['max_ms'], el)
        for h in self._pool:
            _L().rsuper_timer_event_destroy(h)
        self._pool = []
        return out


TIMER = None   # set to a KernelTimer to time conv launches
WEIGHTS_EPOCH = 0   # bumped whenever a HIP kernel rewrites parameters behind autograd's back (fused optimiser / EMA update):
                    # invalidates cached weight fragments (tensor._version does not see raw-pointer writes)
GRAD_DEST = None   # optional callable(param) -> tensor view to write that parameter's gradient into (rsuper_amd.reducer)


def grad_dest(w, shape=None):
    """Destination tensor for the gradient of parameter `w`: a flat-bucket view when a GradReducer is attached, else a fresh
    tensor.  Kernels overwrite it completely."""
    if GRAD_DEST is not None:
        v = GRAD_DEST(w)
        if v is not None:
            return v
    return torch.empty_like(w) if shape is None else torch.empty(shape, device=w.device, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ side stream
# Weight gradients depend only on (x, dY) and are consumed by the optimiser, so they run on a second HIP stream
# concurrently with the data-gradient chain of the main stream (they fill CUs left idle by kernel tails and by the
# small low-resolution layers).  The main stream joins the side stream at the end of backward (autograd callback).
# Opt-in (RSUPER_WGRAD_OVERLAP=1): same-box 1-2 % faster steps (14.10 -> 13.99 ms), but both streams' kernels fill the
# chip, so they mostly time-share it and every per-kernel duration (HIP events, rocprofv3) is inflated by its neighbour;
# the default keeps one stream so kernel timings and the roofline accounting stay clean.
_SIDE = None
_PENDING = []
_CALLBACK_QUEUED = False


def overlap_mode():
    """RSUPER_WGRAD_OVERLAP: '0' never (default), '1' always, 'auto': only where a launch cannot fill the chip.
    Same-box step times of round 2: 12.58 ms ('0'), 12.57 ms ('auto'), 12.72 ms ('1')."""
    return os.environ.get('RSUPER_WGRAD_OVERLAP', '0')


def overlap_enabled(voxels=None):
    """Weight gradients on the side stream?  `auto`: for volumes of at most 24^3 voxels per sample -- there the
    data-gradient and the weight-gradient launches each occupy a fraction of the 256 CUs (72-144 persistent blocks), so the
    two chains run side by side instead of taking turns; at 48^3 / 96^3 every launch fills the chip and concurrency only
    interleaves them.  voxels=None asks whether the side stream is in use at all (the gradient reducer orders its
    collectives after it)."""
    m = overlap_mode()
    if voxels is None and reduce_side_enabled():
        return True
    if m == '1':
        return True
    if m == 'auto':
        return voxels is None or voxels <= 24 ** 3
    return False


def _join_per_block():
    """Under DDP (any world size, incl. the 1-rank `bench.py --force-ddp`) the reducer's hooks copy each gradient into its
    bucket as soon as the block's backward returns, on the main stream: the side stream is joined at the end of every
    block's backward instead of once at the end of the whole backward pass."""
    import torch.distributed as dist
    # with rsuper_amd.reducer.GradReducer attached (GRAD_DEST set) the collectives are ordered on the side stream itself
    return GRAD_DEST is None and dist.is_available() and dist.is_initialized()


def side_stream():
    global _SIDE
    if _SIDE is None:
        _SIDE = torch.cuda.Stream()
    return _SIDE


def join_side():
    """Make the current (main) stream wait for all weight-gradient work queued on the side stream."""
    global _CALLBACK_QUEUED
    cur = torch.cuda.current_stream()
    while _PENDING:
        cur.wait_event(_PENDING.pop())
    _CALLBACK_QUEUED = False


def _queue_join():
    global _CALLBACK_QUEUED
    if not _CALLBACK_QUEUED:
        _CALLBACK_QUEUED = True
        torch.autograd.Variable._execution_engine.queue_callback(join_side)


class _Side:
    """Context: run the enclosed launches on the side stream after `after` (an event on the main stream)."""

    def __init__(self, enabled, tensors):
        self.enabled, self.tensors = enabled, [t for t in tensors if t is not None]

    def __enter__(self):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record()
            self.ctx = torch.cuda.stream(side_stream())
            self.ctx.__enter__()
            side_stream().wait_event(ev)
        return self

    def __exit__(self, *a):
        if self.enabled:
            done = torch.cuda.Event()
            done.record(side_stream())
            _PENDING.append(done)
            self.ctx.__exit__(*a)
            for t in self.tensors:
                t.record_stream(side_stream())     # keep inputs alive until the side stream has consumed them
            if not _join_per_block():
                _queue_join()
        return False


def igemm(epi, a, b, packed, n_cols, bn, dims, out, out_ld=None, res=None, part=None, ea=None, eb=None, out2=None, out_split=0):
    """a, b, ea, eb: Src (b/eb may be None).  res: Src or None.  dims = (N, D, H, W).
    out2 / out_split: columns [out_split, n_cols) go to the tensor `out2` (same row stride as `out`, allocated behind it in the same buffer) --
    rsuper_conv3_igemm_split_out, depth-reuse kernel only."""
    dt = _DT[a.t.dtype]
    N, D, H, W = dims
    ra = (None, 0) if res is None else (_ptr(res.t, res.off), res.ld)

    def run():
        if out2 is not None:
            delta = (out2.data_ptr() - out.data_ptr()) // out.element_size()
            _l.check(_L().rsuper_conv3_igemm_split_out(dt, epi, *a.args(), *(b.args() if b is not None else _NONE), _ptr(packed), n_cols, bn,
                                                       N, D, H, W, _ptr(out), out.shape[-1] if out_ld is None else out_ld, ra[0], ra[1], _ptr(part),
                                                       *(ea.args() if ea is not None else _NONE), *(eb.args() if eb is not None else _NONE),
                                                       out_split, delta, _stream()), 'conv3_igemm_split_out')
            return
        _l.check(_L().rsuper_conv3_igemm(dt, epi, *a.args(), *(b.args() if b is not None else _NONE), _ptr(packed), n_cols, bn,
                                         N, D, H, W, _ptr(out), out.shape[-1] if out_ld is None else out_ld, ra[0], ra[1], _ptr(part),
                                         *(ea.args() if ea is not None else _NONE), *(eb.args() if eb is not None else _NONE),
                                         _stream()), 'conv3_igemm')
    if TIMER is not None:
        K = a.C + (b.C if b is not None else 0)
        TIMER.launch('conv3d_igemm_fwd' if epi == 0 else 'conv3d_igemm_dgrad', 2.0 * N * D * H * W * n_cols * K * 27, run)
    else:
        run()


def igemm_s2(mode, a, b, packed, n_cols, full_dims, out, part, ea=None):
    """Stride-2 forward (mode 1: a = Src with statistics on the full grid, out on the half grid) / its data gradient (mode 2: a, b = dy sources
    on the half grid, out and the forward input `ea` on the full grid) -- csrc/conv3d_igemm_s2.hip, 64-column blocks."""
    dt = _DT[a.t.dtype]
    N, FD, FH, FW = full_dims
    bx = (None, 0, 0) if b is None else (_ptr(b.t, b.off), b.ld, b.C)
    ex = (None, 0, None) if ea is None else (_ptr(ea.t, ea.off), ea.ld, _ptr(ea.mr))

    def run():
        _l.check(_L().rsuper_conv3_igemm_s2(dt, mode, _ptr(a.t, a.off), a.ld, a.C, _ptr(a.mr) if mode == 1 else None, *bx, _ptr(packed), n_cols,
                                            N, FD, FH, FW, _ptr(out), out.shape[-1], _ptr(part), *ex, _stream()), 'conv3_igemm_s2')
    if TIMER is not None:
        K = a.C + (b.C if b is not None else 0)
        OD, OH, OW = (FD + 1) // 2, (FH + 1) // 2, (FW + 1) // 2
        TIMER.launch('conv3d_igemm_fwd' if mode == 1 else 'conv3d_igemm_dgrad', 2.0 * N * OD * OH * OW * n_cols * K * 27, run)
    else:
        run()


def strided_kernel(dtype, dims, direction):
    """Whether a strided [conv1 | shortcut] GEMM takes the parity-class kernel (csrc/conv3d_igemm_s2.hip: the minimal MFMA work, no
    full-resolution temporary) or the rounds-1/2 evaluation (the tuned stride-1 kernels at full resolution + subsample / zero-stuffed dy: 8x
    the work).  Measured on MI355X (tools/bench_conv.py, B = 2, 96^3 / 48^3 / 24^3 inputs): f32 forward 5.3x / 3.4x / 3.8x and data gradient
    3.7x / 3.6x / 4.2x faster; bf16 forward 2.3x / 1.3x / 1.7x, data gradient 2.2x / 1.8x / 1.3x -- so it is the default everywhere;
    RSUPER_S2_KERNEL=0 selects the old evaluation (tests run both, A/B)."""
    if os.environ.get('RSUPER_S2_KERNEL', '1') == '0':
        return False
    if direction == 'wgrad':          # csrc/conv3d_wgrad_s2.hip (round 3); RSUPER_S2_WGRAD=0: stride-1 kernel on the zero-stuffed dy
        return os.environ.get('RSUPER_S2_WGRAD', '1') != '0'
    return True


def reduce_side_enabled():
    """RSUPER_WGRAD_REDUCE_SIDE=1 (opt-in; same-box A/B 11.68 vs 11.24 ms per step: the cross-stream event waits cost more than the
    overlap gains): the slab reduction of every weight gradient (a ~10 us bandwidth-bound kernel that only the
    optimiser waits for) runs on the side stream, concurrently with the data-gradient chain, instead of between two of its kernels."""
    return os.environ.get('RSUPER_WGRAD_REDUCE_SIDE', '0') == '1'


# Deferred slab reductions: the two weight gradients of a BasicBlock's backward (conv2; conv1 + shortcut) only write their per-split slabs, and ONE
# batched launch at the end of the block's backward sums both -- the per-layer reduction is a ~10 us launch at the dependent-launch floor (34 per UNet
# step).  Deferring ALL of them to the end of the backward pass was measured too (one launch of 333 us instead of 34 x 11 us: the 1.5 GB of slabs
# are then read cold from HBM instead of hot from L2 / Infinity Cache -- no gain, and the slabs stay allocated), hence per block.
# RSUPER_WGRAD_DEFER=0 restores the per-layer launches.
_DEFERRED = []
DEFER_WGRAD_REDUCE = os.environ.get('RSUPER_WGRAD_DEFER', '1') == '1'


FUSE_STATS_REDUCE = os.environ.get('RSUPER_FUSE_STATS_REDUCE', '1') == '1'      # =0: every statistics finalisation as its own launch (A/B)


def flush_wgrad_reduces(stats=None):
    """Sum the slabs of every weight gradient whose reduction was deferred (one launch on the current stream).  Idempotent.
    stats: up to two (part, cnt, mode, split) statistics buffers finalised by the SAME launch (rsuper_conv3_wgrad_reduce_batch_stats: the arithmetic of
    stats_finalize, bit-identical); returns the list of their results in stats_finalize's form.  Without deferred reductions to ride on they are finalised
    by their own launches."""
    import ctypes
    if stats and (not _DEFERRED or not FUSE_STATS_REDUCE or len(stats) > 2 or len(_DEFERRED) > 48):
        res = [stats_finalize(p_, cnt, mode=mode, split=split) for p_, cnt, mode, split in stats]
        flush_wgrad_reduces()
        return res
    if not _DEFERRED:
        return [] if stats else None
    ents = list(_DEFERRED)
    del _DEFERRED[:]
    n = len(ents)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    args = (n, PA(*[_ptr(e[0]) for e in ents]), IA(*[e[1] for e in ents]), IA(*[e[2] for e in ents]), IA(*[e[3] for e in ents]),
            IA(*[e[4] for e in ents]), PA(*[_ptr(e[5]) for e in ents]), PA(*[_ptr(e[6]) for e in ents]))
    res = None
    if stats:
        m = len(stats)
        outs = [torch.empty((p_.shape[0], p_.shape[2], 2), device=p_.device, dtype=torch.float32) for p_, _, _, _ in stats]
        PB, IB, DB = ctypes.c_void_p * m, ctypes.c_int * m, ctypes.c_double * m
        sargs = (m, PB(*[_ptr(p_) for p_, _, _, _ in stats]), IB(*[p_.shape[0] for p_, _, _, _ in stats]), IB(*[p_.shape[1] for p_, _, _, _ in stats]),
                 IB(*[p_.shape[2] for p_, _, _, _ in stats]), DB(*[float(c) for _, c, _, _ in stats]), IB(*[mo for _, _, mo, _ in stats]),
                 IB(*[sp for _, _, _, sp in stats]), PB(*[_ptr(o) for o in outs]), EPS)
        res = []
        for (p_, _, _, sp), o in zip(stats, outs):
            N, C = p_.shape[0], p_.shape[2]
            if sp:
                flat = o.view(-1)
                res.append((flat[:N * sp * 2].view(N, sp, 2), flat[N * sp * 2:].view(N, C - sp, 2)))
            else:
                res.append(o)

        def run():
            _l.check(_L().rsuper_conv3_wgrad_reduce_batch_stats(*args, *sargs, _stream()), 'conv3_wgrad_reduce_batch_stats')
    else:
        def run():
            _l.check(_L().rsuper_conv3_wgrad_reduce_batch(*args, _stream()), 'conv3_wgrad_reduce_batch')
    if TIMER is not None:
        TIMER.launch('conv3d_wgrad_reduce', 0.0, run)
    else:
        run()
    return res


def _defer_reduce(ws, splits, Cin_t, Ya, Yb, dwa, dwb):
    _DEFERRED.append((ws, splits, Cin_t, Ya, Yb, dwa, dwb))


def wgrad(xa, xb, ya, yb, dwa, dwb, dims, side_reduce=False, defer=False):
    """side_reduce (only from inside an autograd backward, where the join callback can be queued): partial slabs on the current
    stream, their reduction into dwa / dwb on the side stream.  defer: partial slabs now, the reduction in the batched launch of the caller's
    flush_wgrad_reduces() (before dwa / dwb leave the caller)."""
    dt = _DT[xa.t.dtype]
    N, D, H, W = dims
    Mtot = ya.C + (yb.C if yb is not None else 0)
    splits = _L().rsuper_conv3_wgrad_splits(dt, xa.C, xb.C if xb is not None else 0, ya.C, yb.C if yb is not None else 0, N, D, H, W)
    assert splits >= 1
    yb_args = (None, 0, 0) if yb is None else (_ptr(yb.t, yb.off), yb.ld, yb.C)
    Cin_t = xa.C + (xb.C if xb is not None else 0)
    ws = torch.empty((splits * 27 * Mtot * Cin_t,), device=dwa.device, dtype=torch.float32)
    flops = 2.0 * N * D * H * W * Mtot * Cin_t * 27

    if defer and not side_reduce:
        def run_partial_only():
            _l.check(_L().rsuper_conv3_wgrad_partial(dt, use_tr(), *xa.args(), *(xb.args() if xb is not None else _NONE),
                                                     _ptr(ya.t, ya.off), ya.ld, ya.C, *yb_args, _ptr(ws), N, D, H, W, splits, _stream()),
                     'conv3_wgrad_partial')
        if TIMER is not None:
            TIMER.launch('conv3d_wgrad', flops, run_partial_only)
        else:
            run_partial_only()
        _defer_reduce(ws, splits, Cin_t, ya.C, yb.C if yb is not None else 0, dwa, dwb)
        return
    if not side_reduce:
        def run():
            _l.check(_L().rsuper_conv3_wgrad(dt, use_tr(), *xa.args(), *(xb.args() if xb is not None else _NONE),
                                             _ptr(ya.t, ya.off), ya.ld, ya.C, *yb_args, _ptr(dwa), _ptr(dwb), _ptr(ws), N, D, H, W, splits,
                                             _stream()), 'conv3_wgrad')
        if TIMER is not None:
            TIMER.launch('conv3d_wgrad', flops, run)
        else:
            run()
        return

    def run_partial():
        _l.check(_L().rsuper_conv3_wgrad_partial(dt, use_tr(), *xa.args(), *(xb.args() if xb is not None else _NONE),
                                                 _ptr(ya.t, ya.off), ya.ld, ya.C, *yb_args, _ptr(ws), N, D, H, W, splits, _stream()),
                 'conv3_wgrad_partial')

    def run_reduce():
        _l.check(_L().rsuper_conv3_wgrad_reduce(_ptr(ws), splits, Cin_t, ya.C, yb.C if yb is not None else 0, _ptr(dwa), _ptr(dwb),
                                                _stream()), 'conv3_wgrad_reduce')
    if TIMER is not None:
        TIMER.launch('conv3d_wgrad', flops, run_partial)
    else:
        run_partial()
    with _Side(True, (ws, dwa, dwb)):
        if TIMER is not None:
            TIMER.launch('conv3d_wgrad_reduce', 0.0, run_reduce)      # events on the side stream: counted in the union of intervals
        else:
            run_reduce()


def wgrad_s2(xa, ya, yb, dwa, dwb, full_dims):
    """Weight gradient of the stride-2 [conv1 | shortcut] pair: xa (with statistics) on the full grid, ya / yb = dy sources on the half grid
    (csrc/conv3d_wgrad_s2.hip -- every x voxel staged once, 27 dense taps on the half grid, no zero-stuffed dy)."""
    dt = _DT[xa.t.dtype]
    N, FD, FH, FW = full_dims
    Mtot = ya.C + (yb.C if yb is not None else 0)
    splits = _L().rsuper_conv3_wgrad_s2_splits(dt, xa.C, Mtot, N, FD, FH, FW)
    assert splits >= 1
    yb_args = (None, 0, 0) if yb is None else (_ptr(yb.t, yb.off), yb.ld, yb.C)
    ws = torch.empty((splits * 27 * Mtot * xa.C,), device=dwa.device, dtype=torch.float32)

    def run():
        _l.check(_L().rsuper_conv3_wgrad_s2(dt, _ptr(xa.t, xa.off), xa.ld, xa.C, _ptr(xa.mr), _ptr(ya.t, ya.off), ya.ld, ya.C, *yb_args,
                                            _ptr(dwa), _ptr(dwb), _ptr(ws), N, FD, FH, FW, splits, _stream()), 'conv3_wgrad_s2')
    if TIMER is not None:
        OD, OH, OW = (FD + 1) // 2, (FH + 1) // 2, (FW + 1) // 2
        TIMER.launch('conv3d_wgrad', 2.0 * N * OD * OH * OW * Mtot * xa.C * 27, run)
    else:
        run()


def in_bwd_finalize(g, x, gm, out_C, add1=None):
    """g, x: Src (x carries mr).  gm: (N, C, 2) contiguous.  Returns new tensor (N,D,H,W,out_C)."""
    N, D, H, W = g.t.shape[:4]
    out = torch.empty((N, D, H, W, out_C), device=g.t.device, dtype=g.t.dtype)
    a1 = (None, 0) if add1 is None else (_ptr(add1), add1.shape[-1])
    _l.check(_L().rsuper_in_bwd_finalize(_DT[g.t.dtype], _ptr(g.t, g.off), g.ld, _ptr(x.t, x.off), x.ld, _ptr(x.mr), _ptr(gm),
                                         a1[0], a1[1], None, 0, _ptr(out), out_C, N, D * H * W, out_C, _stream()), 'in_bwd_finalize')
    return out


# ------------------------------------------------------------------------------------------------ BasicBlock
class BasicBlockFn(torch.autograd.Function):
    """BasicBlock.forward (model/dim3/conv_layers.py:86-94) with pre-activation ConvNormAct members:
        out = conv2(relu(IN(conv1(relu(IN(x)))))) + [convS(relu(IN(x))) | x]
    conv1 and the shortcut conv read the same normalised input and run as ONE GEMM with N = 2*Cout."""

    @staticmethod
    def forward(ctx, xa, mra, xb, mrb, w1, w2, ws, packs=None, stride=1):
        """packs: optional (tensors, bns) from pack_weights_batch / block_pack_specs (whole-network batched packing).
        stride 2 (down_block(pool=False), unet_utils.py:38-39): conv1 and the shortcut conv run as one strided GEMM on the parity-class
        kernel (csrc/conv3d_igemm_s2.hip: the minimal MFMA work, forward and data gradient; their weight gradient still reads a
        zero-stuffed dy at full resolution), conv2 at the half resolution."""
        if stride == 2:
            return _BB._forward_s2(ctx, xa, mra, w1, w2, ws)
        assert stride == 1
        ctx.stride = 1
        _chk_act(xa)
        N, D, H, W, Ca = xa.shape
        Cb = 0 if xb is None else xb.shape[-1]
        Cout = w1.shape[0]
        has_sc = ws is not None
        assert w1.shape[1] == Ca + Cb and (has_sc or (Cb == 0 and Ca == Cout))
        dims = (N, D, H, W)
        dev, dt = xa.device, xa.dtype
        tiles = _L().rsuper_conv3_tiles(D, H, W)
        cnt = D * H * W
        sa = Src(xa, mr=mra)
        sb = None if xb is None else Src(xb, mr=mrb)
        # conv1 (+ shortcut): one GEMM
        nc1 = Cout * (2 if has_sc else 1)
        if packs is None:                  # one pack launch per block and direction (fragments stay hot in L2 for the convs below)
            # with gradients wanted, the data-gradient fragments are packed by the same launch (one pack launch per block instead of
            # two; they are read once, much later, so being cold in L2 by then costs nothing measurable)
            both = os.environ.get('RSUPER_PACK_BOTH', '1') == '1' and any(ctx.needs_input_grad)
            packs = block_packs(w1, w2, ws, Ca, Cb, dt, tiles * N, both, dims)
        bn1, wp1 = packs[1][0], packs[0][0]
        part = part_buffer(dt, dims, nc1, bn1, dev)
        # [y1 | shortcut] halves narrower than a 128-byte line (32 bf16 channels: up4.0): two tensors instead of an interleaved one, so that conv2's staging,
        # its weight gradient, the mask of its data gradient and the InstanceNorm-backward tail read whole cache lines (rsuper_conv3_igemm_split_out; the
        # depth-reuse kernel's epilogue only -- other kernels keep the interleaved output)
        split_ys = (has_sc and Cout * xa.element_size() < 128 and Cout % 32 == 0 and os.environ.get('RSUPER_SPLIT_YS', '1') == '1'
                    and _L().rsuper_conv3_kd_bn(_DT[dt], 0, N, D, H, W, nc1, 0) == bn1)
        if split_ys:
            ys2 = torch.empty((2, N, D, H, W, Cout), device=dev, dtype=dt)
            ys, sc_t = ys2[0], ys2[1]
            igemm(0, sa, sb, wp1, nc1, bn1, dims, ys, part=part, out2=sc_t, out_split=Cout)
        else:
            ys = torch.empty((N, D, H, W, nc1), device=dev, dtype=dt)
            igemm(0, sa, sb, wp1, nc1, bn1, dims, ys, part=part)
        mr_y1 = stats_finalize(part, cnt, split=Cout)[0] if has_sc else stats_finalize(part, cnt)
        # conv2 + residual
        bn2, wp2 = packs[1][1], packs[0][1]
        out = torch.empty((N, D, H, W, Cout), device=dev, dtype=dt)
        part2 = part_buffer(dt, dims, Cout, bn2, dev)
        res = (Src(sc_t) if split_ys else Src(ys, C=Cout, off=Cout)) if has_sc else Src(xa)
        igemm(0, Src(ys, C=Cout, mr=mr_y1), None, wp2, Cout, bn2, dims, out, res=res, part=part2)
        mr_out = stats_finalize(part2, cnt)
        ctx.save_for_backward(xa, mra, xb, mrb, ys, mr_y1, w1, w2, ws)
        ctx.packs = packs if (packs is not None and len(packs[0]) >= 4) else None
        ctx.mark_non_differentiable(mr_out)
        ctx.set_materialize_grads(False)        # no zero-fill kernel for the statistics output's (never used) gradient
        return out, mr_out

    @staticmethod
    def _forward_s2(ctx, xa, mra, w1, w2, ws):
        _chk_act(xa)
        assert ws is not None, 'a strided BasicBlock always has a convolutional shortcut (conv_layers.py:82-84)'
        N, D, H, W, Ca = xa.shape
        Cout = w1.shape[0]
        dims = (N, D, H, W)
        dev, dt = xa.device, xa.dtype
        tiles = _L().rsuper_conv3_tiles(D, H, W)
        sa = Src(xa, mr=mra)
        nc1 = 2 * Cout
        if not strided_kernel(dt, dims, 'fwd'):
            specs, bns = block_pack_specs(w1, w2, ws, Ca, 0, dt, tiles * N, False, dims)
            wp = pack_weights_batch(dt, specs)
            full = torch.empty((N, D, H, W, nc1), device=dev, dtype=dt)
            igemm(0, sa, None, wp[0], nc1, bns[0], dims, full)                   # [conv1 | shortcut] at stride 1, no statistics
            ys, mr_ys = subsample2(full)
            del full
            OD, OH, OW = ys.shape[1:4]
            mr_y1 = mr_ys[:, :Cout].contiguous()
        else:
            # [conv1 | shortcut] as ONE strided GEMM on the parity-class kernel: 1/8 of the MFMA work of the full-resolution evaluation
            OD, OH, OW = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
            wp1 = pack_weights(dt, 0, w1, ws, Ca, 0, Cout, Cout, 64)
            ys = torch.empty((N, OD, OH, OW, nc1), device=dev, dtype=dt)
            part1 = torch.empty((N, _L().rsuper_conv3_s2_part_rows(_DT[dt], 1, Ca, 0, nc1, N, D, H, W), nc1, 2), device=dev, dtype=torch.float32)
            igemm_s2(1, sa, None, wp1, nc1, dims, ys, part1)
            mr_y1 = stats_finalize(part1, OD * OH * OW, split=Cout)[0]
        dims2 = (N, OD, OH, OW)
        tiles2 = _L().rsuper_conv3_tiles(OD, OH, OW)
        bn2 = pick_bn(Cout, dt, tiles2 * N, dims2)
        wp2 = pack_weights(dt, 0, w2, None, Cout, 0, Cout, 0, bn2)
        out = torch.empty((N, OD, OH, OW, Cout), device=dev, dtype=dt)
        part2 = part_buffer(dt, dims2, Cout, bn2, dev)
        igemm(0, Src(ys, C=Cout, mr=mr_y1), None, wp2, Cout, bn2, dims2, out, res=Src(ys, C=Cout, off=Cout), part=part2)
        mr_out = stats_finalize(part2, OD * OH * OW)
        ctx.save_for_backward(xa, mra, None, None, ys, mr_y1, w1, w2, ws)
        ctx.packs = None
        ctx.stride = 2
        ctx.mark_non_differentiable(mr_out)
        ctx.set_materialize_grads(False)
        return out, mr_out

    @staticmethod
    def _backward_s2(ctx, dout):
        xa, mra, _, _, ys, mr_y1, w1, w2, ws = ctx.saved_tensors
        dout = dout.contiguous()
        N, D, H, W, Ca = xa.shape
        Cout = w1.shape[0]
        OD, OH, OW = ys.shape[1:4]
        dims, dims2 = (N, D, H, W), (N, OD, OH, OW)
        dev, dt = xa.device, xa.dtype
        y1 = Src(ys, C=Cout, mr=mr_y1)
        sdo = Src(dout)
        # conv2 at the half resolution: as in the stride-1 block
        tiles2 = _L().rsuper_conv3_tiles(OD, OH, OW)
        bn = pick_bn(Cout, dt, tiles2 * N, dims2)
        wpd2 = pack_weights(dt, 1, w2, None, Cout, 0, Cout, 0, bn)
        g1 = torch.empty((N, OD, OH, OW, Cout), device=dev, dtype=dt)
        part = part_buffer(dt, dims2, Cout, bn, dev, epi=1)
        igemm(1, sdo, None, wpd2, Cout, bn, dims2, g1, part=part, ea=y1)
        gm1 = stats_finalize(part, OD * OH * OW, mode=1)
        dw2 = grad_dest(w2)
        wgrad(y1, None, sdo, None, dw2, None, dims2)
        dy1 = in_bwd_finalize(Src(g1), y1, gm1, Cout)
        # conv1 + shortcut.  Parity-class / strided kernels read [dy1 | dOut] on the half grid; the rounds-1/2 evaluation needs the output
        # gradient zero-stuffed to the even voxels of the full-resolution grid
        sa = Src(xa, mr=mra)
        tiles = _L().rsuper_conv3_tiles(D, H, W)
        g0 = torch.empty((N, D, H, W, Ca), device=dev, dtype=dt)
        k_dgrad, k_wgrad = strided_kernel(dt, dims, 'dgrad'), strided_kernel(dt, dims, 'wgrad')
        dfull = None
        if not (k_dgrad and k_wgrad):
            dfull = torch.empty((N, D, H, W, 2 * Cout), device=dev, dtype=dt)
            subsample2_scatter(dy1, dfull, 0, dims)
            subsample2_scatter(dout, dfull, Cout, dims)
        if not k_dgrad:
            bnd = pick_bn(Ca, dt, tiles * N, dims)
            wpd1 = pack_weights(dt, 1, w1, ws, Cout, Cout, Ca, 0, bnd)
            part0 = part_buffer(dt, dims, Ca, bnd, dev, epi=1)
            igemm(1, Src(dfull, C=Cout), Src(dfull, C=Cout, off=Cout), wpd1, Ca, bnd, dims, g0, part=part0, ea=sa)
        else:
            # the transposed convolution by output parity classes, straight from [dy1 | dOut] on the half grid (no zero-stuffed operand)
            wpd1 = pack_weights(dt, 1, w1, ws, Cout, Cout, Ca, 0, 64)
            part0 = torch.empty((N, _L().rsuper_conv3_s2_part_rows(_DT[dt], 2, Cout, Cout, Ca, N, D, H, W), Ca, 2), device=dev, dtype=torch.float32)
            igemm_s2(2, Src(dy1), sdo, wpd1, Ca, dims, g0, part0, ea=sa)
        gm0 = stats_finalize(part0, D * H * W, mode=1)
        dw1, dws = grad_dest(w1), grad_dest(ws)
        if k_wgrad:
            wgrad_s2(sa, Src(dy1), sdo, dw1, dws, dims)
        else:
            wgrad(sa, None, Src(dfull, C=Cout), Src(dfull, C=Cout, off=Cout), dw1, dws, dims)
        dxa = in_bwd_finalize(Src(g0), sa, gm0, Ca)
        return dxa, None, None, None, dw1, dw2, dws, None, None

    @staticmethod
    def backward(ctx, dout, _unused):
        if ctx.stride == 2:
            return _BB._backward_s2(ctx, dout)
        try:
            return _BB._backward_s1(ctx, dout)
        except BaseException:
            # an exception between the first wgrad(defer=True) and flush_wgrad_reduces() (e.g. the OutOfMemoryError graph._verify survives) must not
            # leave entries behind: they pin the split slabs and would be reduced into orphaned buffers by the next backward
            del _DEFERRED[:]
            raise

    @staticmethod
    def _backward_s1(ctx, dout):
        xa, mra, xb, mrb, ys, mr_y1, w1, w2, ws = ctx.saved_tensors
        dout = dout.contiguous()
        N, D, H, W, Ca = xa.shape
        Cb = 0 if xb is None else xb.shape[-1]
        Cin, Cout = Ca + Cb, w1.shape[0]
        has_sc = ws is not None
        dims = (N, D, H, W)
        dev, dt = xa.device, xa.dtype
        tiles = _L().rsuper_conv3_tiles(D, H, W)
        cnt = D * H * W
        y1 = Src(ys, C=Cout, mr=mr_y1)
        sdo = Src(dout)
        # conv2: data gradient (ReLU mask + IN sums fused), weight gradient
        if ctx.packs is not None:
            bpk = (ctx.packs[0][2:], ctx.packs[1][2:])
        else:
            bufs, bbns = block_packs(w1, w2, ws, Ca, Cb, dt, tiles * N, True, dims)
            bpk = (bufs[2:], bbns[2:])
        bn, wpd2 = bpk[1][0], bpk[0][0]
        g1 = torch.empty((N, D, H, W, Cout), device=dev, dtype=dt)
        part = part_buffer(dt, dims, Cout, bn, dev, epi=1)
        igemm(1, sdo, None, wpd2, Cout, bn, dims, g1, part=part, ea=y1)
        gm1 = stats_finalize(part, cnt, mode=1)
        # the join with the side stream is deferred to the end of backward: only safe when AccumulateGrad merely stores
        # the new gradient (p.grad is None, zero_grad(set_to_none=True)); an in-place `p.grad += dw` would race
        fresh = all(w.grad is None for w in (w1, w2, ws) if w is not None)
        ov = overlap_enabled(D * H * W) and fresh
        sr = reduce_side_enabled() and fresh and not ov
        dw2 = grad_dest(w2)
        with _Side(ov, (ys, mr_y1, dout, dw2)):
            wgrad(y1, None, sdo, None, dw2, None, dims, side_reduce=sr, defer=DEFER_WGRAD_REDUCE and not ov)
        dy1 = in_bwd_finalize(Src(g1), y1, gm1, Cout)
        # conv1 (+ shortcut): fused data gradient over [dY1 | dOut], fused weight gradient
        sa = Src(xa, mr=mra)
        sb = None if xb is None else Src(xb, mr=mrb)
        # one gradient tensor per forward source when each source has its own launch (a 32-channel slice of a 96-channel row is 64 of every 192 bytes: the
        # InstanceNorm-backward tails then fetch half-used cache lines -- up4.0: 80 / 145 us for the two tails against 61 / 123 us on contiguous tensors)
        split_g0 = len(bpk[0]) == 3 and os.environ.get('RSUPER_SPLIT_G0', '1') == '1'
        g0 = None if split_g0 else torch.empty((N, D, H, W, Cin), device=dev, dtype=dt)
        g0s = [torch.empty((N, D, H, W, Ca), device=dev, dtype=dt), torch.empty((N, D, H, W, Cb), device=dev, dtype=dt)] if split_g0 else None
        # With the slab reductions deferred to one launch per block, that launch also finalises the InstanceNorm-backward rows of this data gradient
        # (flush_wgrad_reduces(stats=...)): it then runs BEFORE the InstanceNorm-backward tail instead of after it.
        fuse = FUSE_STATS_REDUCE and DEFER_WGRAD_REDUCE and not ov and not sr
        pending = []
        if len(bpk[0]) == 3:                  # one launch per forward source (block_pack_specs / split_dgrad_sources)
            gm0 = []
            for si, (c0, cn, src, wp_, bn) in enumerate(((0, Ca, sa, bpk[0][1], bpk[1][1]), (Ca, Cb, sb, bpk[0][2], bpk[1][2]))):
                part0 = part_buffer(dt, dims, cn, bn, dev, epi=1)
                if split_g0:
                    igemm(1, Src(dy1), sdo if has_sc else None, wp_, cn, bn, dims, g0s[si], part=part0, ea=src)
                else:
                    igemm(1, Src(dy1), sdo if has_sc else None, wp_, cn, bn, dims, g0[..., c0:], out_ld=Cin, part=part0, ea=src)
                if fuse:
                    pending.append((part0, cnt, 1, 0))
                else:
                    gm0.append(stats_finalize(part0, cnt, mode=1))
        else:
            bn, wpd1 = bpk[1][1], bpk[0][1]
            part0 = part_buffer(dt, dims, Cin, bn, dev, epi=1)
            igemm(1, Src(dy1), sdo if has_sc else None, wpd1, Cin, bn, dims, g0, part=part0, ea=sa, eb=sb)
            if fuse:
                pending.append((part0, cnt, 1, 0 if xb is None else Ca))
            else:
                gm0 = stats_finalize(part0, cnt, mode=1, split=0 if xb is None else Ca)
        dw1 = grad_dest(w1)
        dws = grad_dest(ws) if has_sc else None
        with _Side(ov, (xa, mra, xb, mrb, dy1, dout, dw1, dws)):
            wgrad(sa, sb, Src(dy1), sdo if has_sc else None, dw1, dws, dims, side_reduce=sr, defer=DEFER_WGRAD_REDUCE and not ov)
        if fuse:
            fin = flush_wgrad_reduces(stats=pending)
            gm0 = fin if len(bpk[0]) == 3 else fin[0]
        if xb is None:
            dxa = in_bwd_finalize(Src(g0), sa, gm0, Ca, add1=None if has_sc else dout)
            dxb = None
        elif split_g0:
            dxa = in_bwd_finalize(Src(g0s[0]), sa, gm0[0], Ca)
            dxb = in_bwd_finalize(Src(g0s[1]), sb, gm0[1], Cb)
        else:
            dxa = in_bwd_finalize(Src(g0, C=Ca), sa, gm0[0], Ca)
            dxb = in_bwd_finalize(Src(g0, C=Cb, off=Ca), sb, gm0[1], Cb)
        flush_wgrad_reduces()                 # both weight gradients of the block: one reduction launch, slabs still hot in L2
        if (ov or sr) and _join_per_block():
            join_side()
        return dxa, None, dxb, None, dw1, dw2, dws, None, None


_BB = BasicBlockFn        # the class itself: `BasicBlockFn` is rebound to the registered dispatcher op below


# ------------------------------------------------------------------------------------------------ pool / upsample
def _stat_blocks(ovox):
    return max(1, min(2048, ovox // 64))


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool3d(2) (model/dim3/unet_utils.py:35-37) + statistics of the pooled tensor."""

    @staticmethod
    def forward(ctx, x):
        _chk_act(x)
        N, D, H, W, C = x.shape
        OD, OH, OW = D // 2, H // 2, W // 2
        y = torch.empty((N, OD, OH, OW, C), device=x.device, dtype=x.dtype)
        blocks = _stat_blocks(OD * OH * OW)
        part = torch.empty((N, blocks, C, 2), device=x.device, dtype=torch.float32)
        _l.check(_L().rsuper_maxpool2_fwd(_DT[x.dtype], _ptr(x), C, _ptr(y), C, _ptr(part), blocks, N, D, H, W, C, _stream()), 'maxpool2_fwd')
        mr = stats_finalize(part, OD * OH * OW)
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(mr)
        ctx.set_materialize_grads(False)
        return y, mr

    @staticmethod
    def backward(ctx, dy, _unused):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        N, D, H, W, C = x.shape
        dx = torch.empty_like(x)
        _l.check(_L().rsuper_maxpool2_bwd(_DT[x.dtype], _ptr(x), C, _ptr(dy), C, _ptr(dx), C, N, D, H, W, C, _stream()), 'maxpool2_bwd')
        return dx


_MaxPoolFn = MaxPoolFn          # the class itself (hip/library.py rebinds `MaxPoolFn` to the dispatcher entry)


class MaxPoolSkipFn(torch.autograd.Function):
    """nn.MaxPool3d(2) of a tensor that is also a skip connection (model/dim3/unet.py:57-66: x_k feeds down_k and up_k): returns (pooled, statistics,
    x again).  The third output stands for x on the skip path, so that BOTH gradients of x arrive in this node's backward and are summed inside the
    max-pool backward kernel (rsuper_maxpool2_bwd_add) instead of by a separate accumulation launch (4 x 19 us per UNet step)."""

    @staticmethod
    def forward(ctx, x):
        y, mr = _MaxPoolFn.forward(ctx, x)
        return y, mr, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, _unused, dskip):
        if dy is None:
            return dskip
        if dskip is None:
            return _MaxPoolFn.backward(ctx, dy, None)
        (x,) = ctx.saved_tensors
        N, D, H, W, C = x.shape
        dy, dskip = dy.contiguous(), dskip.contiguous()
        dx = torch.empty_like(x)
        rc = _L().rsuper_maxpool2_bwd_add(_DT[x.dtype], _ptr(x), C, _ptr(dy), C, _ptr(dskip), C, _ptr(dx), C, N, D, H, W, C, _stream())
        if rc == 3:                                              # odd sizes: the two steps
            return _MaxPoolFn.backward(ctx, dy, None) + dskip
        _l.check(rc, 'maxpool2_bwd_add')
        return dx


def subsample2(x, with_stats=True):
    """y = x[:, ::2, ::2, ::2] (channels-last) + the InstanceNorm statistics of y (stride-2 evaluation of a stride-1 convolution)."""
    N, D, H, W, C = x.shape
    OD, OH, OW = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    y = torch.empty((N, OD, OH, OW, C), device=x.device, dtype=x.dtype)
    blocks = _stat_blocks(OD * OH * OW)
    part = torch.empty((N, blocks, C, 2), device=x.device, dtype=torch.float32) if with_stats else None
    _l.check(_L().rsuper_subsample2_fwd(_DT[x.dtype], _ptr(x), C, _ptr(y), C, _ptr(part), blocks, N, D, H, W, C, _stream()), 'subsample2_fwd')
    return y, (stats_finalize(part, OD * OH * OW) if with_stats else None)


def subsample2_scatter(dy, out, off, dims):
    """out[..., off:off+C] at the even voxels of the (N, D, H, W) grid = dy, zero elsewhere."""
    N, D, H, W = dims
    C = dy.shape[-1]
    _l.check(_L().rsuper_subsample2_bwd(_DT[dy.dtype], _ptr(dy), C, _ptr(out, off), out.shape[-1], N, D, H, W, C, _stream()), 'subsample2_bwd')


class UpsampleFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='trilinear', align_corners=True) (model/dim3/unet_utils.py:69)."""

    @staticmethod
    def forward(ctx, x, size):
        _chk_act(x)
        N, ID, IH, IW, C = x.shape
        OD, OH, OW = size
        y = torch.empty((N, OD, OH, OW, C), device=x.device, dtype=x.dtype)
        blocks = _stat_blocks(OD * OH * OW)
        part = torch.empty((N, blocks, C, 2), device=x.device, dtype=torch.float32)
        _l.check(_L().rsuper_upsample_fwd(_DT[x.dtype], _ptr(x), C, _ptr(y), C, _ptr(part), blocks, N, ID, IH, IW, OD, OH, OW, C, _stream()),
                  'upsample_fwd')
        mr = stats_finalize(part, OD * OH * OW)
        ctx.in_shape = tuple(x.shape)
        ctx.mark_non_differentiable(mr)
        ctx.set_materialize_grads(False)
        return y, mr

    @staticmethod
    def backward(ctx, dy, _unused):
        dy = dy.contiguous()
        N, ID, IH, IW, C = ctx.in_shape
        OD, OH, OW = dy.shape[1:4]
        dx = torch.empty(ctx.in_shape, device=dy.device, dtype=dy.dtype)
        _l.check(_L().rsuper_upsample_bwd(_DT[dy.dtype], _ptr(dy), C, _ptr(dx), C, N, ID, IH, IW, OD, OH, OW, C, _stream()), 'upsample_bwd')
        return dx, None


# ------------------------------------------------------------------------------------------------ stem / head
class StemFn(torch.autograd.Function):
    """inconv.conv1 = nn.Conv3d(1, C, 3, padding=1, bias=False) (model/dim3/unet_utils.py:14): raw conv, no norm/act."""

    @staticmethod
    def forward(ctx, img, w, dtype):
        assert img.dim() == 5 and img.shape[1] == 1, 'hot path supports in_chan == 1 (reference configs use 1)'
        img = img.contiguous().float()
        N, _, D, H, W = img.shape
        C = w.shape[0]
        y = torch.empty((N, D, H, W, C), device=img.device, dtype=dtype)
        blocks = -(-(D * H * W) // 256)
        part = torch.empty((N, blocks, C, 2), device=img.device, dtype=torch.float32)
        _l.check(_L().rsuper_stem_fwd(_DT[dtype], _ptr(img), _ptr(w), _ptr(y), C, _ptr(part), N, D, H, W, C, _stream()), 'stem_fwd')
        mr = stats_finalize(part, D * H * W)
        ctx.save_for_backward(img, w)
        ctx.mark_non_differentiable(mr)
        ctx.set_materialize_grads(False)
        return y, mr

    @staticmethod
    def backward(ctx, dy, _unused):
        img, w = ctx.saved_tensors
        dy = dy.contiguous()
        N, _, D, H, W = img.shape
        C = w.shape[0]
        dw = grad_dest(w)
        _l.check(_L().rsuper_stem_wgrad(_DT[dy.dtype], _ptr(img), _ptr(dy), C, _ptr(dw), N, D, H, W, C, _stream()), 'stem_wgrad')
        return None, dw, None


class HeadFn(torch.autograd.Function):
    """outc = nn.Conv3d(C, K, kernel_size=1) with bias (model/dim3/unet.py:47) -> logits (N, K, D, H, W) f32."""

    @staticmethod
    def forward(ctx, x, w, b):
        _chk_act(x)
        N, D, H, W, C = x.shape
        K = w.shape[0]
        logits = torch.empty((N, K, D, H, W), device=x.device, dtype=torch.float32)
        _l.check(_L().rsuper_head_fwd(_DT[x.dtype], _ptr(x), C, _ptr(w), _ptr(b), _ptr(logits), N, D * H * W, C, K, _stream()), 'head_fwd')
        ctx.save_for_backward(x, w, b)
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, w, b = ctx.saved_tensors
        dl = dl.contiguous().float()
        N, D, H, W, C = x.shape
        K = w.shape[0]
        dx = torch.empty_like(x)
        dw = grad_dest(w)
        db = grad_dest(b)
        st = _stream()
        _l.check(_L().rsuper_head_bwd(_DT[x.dtype], _ptr(x), C, _ptr(dl), _ptr(w), _ptr(dx), C, _ptr(dw), _ptr(db), N, D * H * W, C, K, st), 'head_bwd')
        return dx, dw, db


class Conv3Fn(torch.autograd.Function):
    """Conv3d(Cin, Cout, 3, padding=1, bias=False) on a RAW channels-last input (no InstanceNorm / ReLU prologue): implicit-GEMM forward,
    the same kernel with mirrored taps for the data gradient, MFMA weight gradient.  Used by MedFormer's SemanticMapGeneration
    (model/dim3/medformer_utils.py:206-236).  Cin and Cout multiples of 8."""

    @staticmethod
    def forward(ctx, x, w):
        _chk_act(x)
        N, D, H, W, Cin = x.shape
        Cout = w.shape[0]
        assert tuple(w.shape[1:]) == (Cin, 3, 3, 3) and Cout % 8 == 0 and w.dtype == torch.float32
        dt, dims = x.dtype, (N, D, H, W)
        tiles = _L().rsuper_conv3_tiles(D, H, W) * N
        bn = pick_bn(Cout, dt, tiles, dims)
        wc = w.contiguous()
        out = torch.empty((N, D, H, W, Cout), device=x.device, dtype=dt)
        igemm(0, Src(x), None, pack_weights(dt, 0, wc, None, Cin, 0, Cout, 0, bn), Cout, bn, dims, out)
        ctx.save_for_backward(x, wc)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        N, D, H, W, Cin = x.shape
        Cout = w.shape[0]
        dt, dims = x.dtype, (N, D, H, W)
        dout = dout.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            bn = pick_bn(Cin, dt, _L().rsuper_conv3_tiles(D, H, W) * N, dims)
            dx = torch.empty_like(x)
            igemm(0, Src(dout), None, pack_weights(dt, 1, w, None, Cout, 0, Cin, 0, bn), Cin, bn, dims, dx)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            wgrad(Src(x), None, Src(dout), None, dw, None, dims)
        return dx, dw


class PlanarFn(torch.autograd.Function):
    """(N, D, H, W, C) channels-last f32, first K channels real -> (N, K, D, H, W) contiguous, and the padded channels-last gradient
    back (csrc/instnorm.hip cl_planar_kernel) -- `x[..., :K].permute(0, 4, 1, 2, 3).contiguous()` in one pass each way."""

    @staticmethod
    def forward(ctx, x, K):
        x = x.contiguous()
        assert x.is_cuda and x.dim() == 5 and x.dtype == torch.float32
        N, D, H, W, C = x.shape
        y = torch.empty((N, K, D, H, W), device=x.device, dtype=torch.float32)
        _l.check(_L().rsuper_cl_planar(_ptr(x), _ptr(y), N, D * H * W, C, K, 0, _stream()), 'cl_planar')
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        N, K, D, H, W = dy.shape
        dx = torch.empty((N, D, H, W, ctx.C), device=dy.device, dtype=torch.float32)
        _l.check(_L().rsuper_cl_planar(_ptr(dy), _ptr(dx), N, D * H * W, ctx.C, K, 1, _stream()), 'cl_planar_bwd')
        return dx, None


class SqueezeExciteFn(torch.autograd.Function):
    """SEBlock (model/dim3/conv_layers.py:159-174) on a channels-last fp32 tensor: y = x * sigmoid(W2 relu(W1 mean(x) + b1) + b2).
    ONE host call per direction (csrc/instnorm.hip: channel statistics + finalize, the excitation of the (N, C) mean vector in a block per
    sample, the per-(sample, channel) affine apply; backward adds the weight-gradient kernel).  The first version kept the excitation
    in ATen with a nested autograd call: 110 us of host time forward and ~200 us backward per block (18 blocks per MedFormer step)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = x.contiguous()
        assert x.is_cuda and x.dim() == 5 and x.dtype == torch.float32 and x.shape[-1] % 4 == 0
        N, C = x.shape[0], x.shape[-1]
        vox = x.shape[1] * x.shape[2] * x.shape[3]
        r = w1.shape[0]
        w1c, w2c = w1.contiguous(), w2.contiguous()
        dev = x.device
        part = torch.empty((N, _L().rsuper_cnorm_rows(vox), C, 2), device=dev, dtype=torch.float32)
        ms = torch.empty((N, C, 2), device=dev, dtype=torch.float32)
        tab = torch.empty((N, C, 2), device=dev, dtype=torch.float32)
        hbuf = torch.empty((N, r), device=dev, dtype=torch.float32)
        y = torch.empty_like(x)
        _l.check(_L().rsuper_se_forward(_ptr(x), _ptr(w1c), _ptr(b1), _ptr(w2c), _ptr(b2), _ptr(part), _ptr(ms), _ptr(tab), _ptr(hbuf), _ptr(y),
                                        N, vox, C, r, _stream()), 'se_forward')
        ctx.save_for_backward(x, w1c, w2c, ms, tab, hbuf)
        ctx.wshapes = (tuple(w1.shape), tuple(w2.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, ms, tab, hbuf = ctx.saved_tensors
        dy = dy.contiguous()
        N, C = x.shape[0], x.shape[-1]
        vox = x.shape[1] * x.shape[2] * x.shape[3]
        r = w1.shape[0]
        dev = x.device
        part = torch.empty((N, _L().rsuper_cnorm_rows(vox), C, 2), device=dev, dtype=torch.float32)
        scratch = torch.empty(N * C * 4 + N * r, device=dev, dtype=torch.float32)          # gm | tab2 | dz1
        gm, tab2, dz1 = scratch[:N * C * 2], scratch[N * C * 2:N * C * 4], scratch[N * C * 4:]
        dx = torch.empty_like(x)
        dw1, dw2 = torch.empty(ctx.wshapes[0], device=dev, dtype=torch.float32), torch.empty(ctx.wshapes[1], device=dev, dtype=torch.float32)
        db1, db2 = torch.empty(r, device=dev, dtype=torch.float32), torch.empty(C, device=dev, dtype=torch.float32)
        _l.check(_L().rsuper_se_backward(_ptr(x), _ptr(dy), _ptr(_se_identity(N, C, dev)), _ptr(w1), _ptr(w2), _ptr(ms), _ptr(tab), _ptr(hbuf),
                                         _ptr(part), _ptr(gm), _ptr(dz1), _ptr(tab2), _ptr(dx), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                                         N, vox, C, r, _stream()), 'se_backward')
        return dx, dw1, db1, dw2, db2


_SE_IDENT = {}


def _se_identity(N, C, device):
    k = (N, C, str(device))
    if k not in _SE_IDENT:
        t = torch.zeros((N, C, 2), device=device, dtype=torch.float32)
        t[..., 1] = 1.0
        _SE_IDENT[k] = t
    return _SE_IDENT[k]


def battn_supported(T, dim_head, heads):
    return bool(_L().rsuper_battn_supported(int(T), int(dim_head), int(heads)))


class BidirAttnFn(torch.autograd.Function):
    """Score / soft-max / mixing core of MedFormer's BidirectionAttention (model/dim3/medformer_utils.py:13-99) in two launches per
    direction (csrc/battn.hip).  fqv (B, L, 2 * inner), mqv (B, T, 2 * inner) f32: q | v halves, channel = dim_head_index * heads + head.
    Returns f_out (B, L, inner), m_out (B, T, inner) in the same channel order."""

    @staticmethod
    def forward(ctx, fqv, mqv, heads, scale):
        fqv, mqv = fqv.contiguous(), mqv.contiguous()
        assert fqv.dtype == torch.float32 and mqv.dtype == torch.float32
        B, L, c2 = fqv.shape
        T = mqv.shape[1]
        inner = c2 // 2
        dh = inner // heads
        dev = fqv.device
        n = _L().rsuper_battn_chunks(L, heads)
        f_out = torch.empty((B, L, inner), device=dev, dtype=torch.float32)
        m_out = torch.empty((B, T, inner), device=dev, dtype=torch.float32)
        lse = torch.empty((B, heads, T, 2), device=dev, dtype=torch.float32)
        part = torch.empty(B * n * T * inner + B * n * heads * T * 2, device=dev, dtype=torch.float32)
        pms = part[B * n * T * inner:]
        _l.check(_L().rsuper_battn_fwd(_ptr(fqv), _ptr(mqv), _ptr(f_out), _ptr(m_out), _ptr(lse), _ptr(part), _ptr(pms), B, L, T, heads, dh,
                                       float(scale), _stream()), 'battn_fwd')
        ctx.save_for_backward(fqv, mqv, m_out, lse)
        ctx.heads, ctx.scale = heads, float(scale)
        return f_out, m_out

    @staticmethod
    def backward(ctx, df, dm):
        fqv, mqv, m_out, lse = ctx.saved_tensors
        B, L, c2 = fqv.shape
        T, heads = mqv.shape[1], ctx.heads
        inner = c2 // 2
        df, dm = df.contiguous(), dm.contiguous()
        n = _L().rsuper_battn_chunks(L, heads)
        d_fqv, d_mqv = torch.empty_like(fqv), torch.empty_like(mqv)
        part = torch.empty(B * n * T * c2, device=fqv.device, dtype=torch.float32)
        _l.check(_L().rsuper_battn_bwd(_ptr(fqv), _ptr(mqv), _ptr(m_out), _ptr(lse), _ptr(df), _ptr(dm), _ptr(d_fqv), _ptr(d_mqv), _ptr(part),
                                       B, L, T, heads, inner // heads, ctx.scale, _stream()), 'battn_bwd')
        return d_fqv, d_mqv, None, None


def token_attn_supported(L, dim_head):
    return bool(_L().rsuper_token_attn_supported(int(L), int(dim_head)))


class TokenAttnFn(torch.autograd.Function):
    """softmax(q k^T * scale) v of the SemanticMapFusion transformer's Attention (model/dim3/trans_layers.py:52-84) in one launch per direction
    (csrc/token_attn.hip).  qkv (B, L, 3 * heads * dim_head) f32 as `to_qkv(x)` produces it; returns (B, L, heads * dim_head)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        if not qkv.is_cuda:
            raise _l.RSuperHipError('TokenAttnFn needs a device tensor (no CPU fallback)')
        qkv = qkv.contiguous()
        assert qkv.dim() == 3 and qkv.dtype == torch.float32 and qkv.shape[-1] % (3 * heads) == 0, (tuple(qkv.shape), qkv.dtype)
        B, L, c3 = qkv.shape
        dh = c3 // (3 * heads)
        o = torch.empty((B, L, heads * dh), device=qkv.device, dtype=torch.float32)
        p = torch.empty((B, heads, L, L), device=qkv.device, dtype=torch.float32)
        _l.check(_L().rsuper_token_attn_fwd(_ptr(qkv), _ptr(o), _ptr(p), B, L, heads, dh, float(scale), _stream()), 'token_attn_fwd')
        ctx.save_for_backward(qkv, p, o)
        ctx.heads, ctx.scale = heads, float(scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, p, o = ctx.saved_tensors
        B, L, c3 = qkv.shape
        do = do.contiguous()
        d_qkv = torch.empty_like(qkv)
        _l.check(_L().rsuper_token_attn_bwd(_ptr(qkv), _ptr(o), _ptr(p), _ptr(do), _ptr(d_qkv), B, L, ctx.heads, c3 // (3 * ctx.heads), ctx.scale, _stream()),
                 'token_attn_bwd')
        return d_qkv, None, None


def channel_stats(x, eps):
    """(mean, rstd) per (sample, channel) of a channels-last fp32 tensor -> (N, C, 2) f32 (csrc/instnorm.hip + stats_finalize)."""
    N, C = x.shape[0], x.shape[-1]
    vox = x.numel() // (N * C)
    rows = _L().rsuper_cnorm_rows(vox)
    part = torch.empty((N, rows, C, 2), device=x.device, dtype=torch.float32)
    mr = torch.empty((N, C, 2), device=x.device, dtype=torch.float32)
    st = _stream()
    _l.check(_L().rsuper_cnorm_stats(_ptr(x), None, None, _ptr(part), N, vox, C, 0, 0, st), 'cnorm_stats')
    _l.check(_L().rsuper_stats_finalize(_ptr(part), N, rows, C, float(vox), float(eps), 0, 0, _ptr(mr), st), 'stats_finalize')
    return mr


CNORM_SMALL_VOX = int(os.environ.get('RSUPER_CNORM_SMALL_VOX', '512'))


class ChannelNormFn(torch.autograd.Function):
    """InstanceNorm3d(affine=False, eps) [+ ReLU] of a channels-last fp32 tensor (N, D, H, W, C), forward and backward on
    csrc/instnorm.hip (per-block partial sums + the deterministic f64 finalize shared with the conv epilogues)."""

    @staticmethod
    def forward(ctx, x, eps, relu):
        if not x.is_cuda:
            raise _l.RSuperHipError('ChannelNormFn needs a device tensor (no CPU fallback)')
        x = x.contiguous()
        assert x.dim() == 5 and x.dtype == torch.float32 and x.shape[-1] % 4 == 0, (tuple(x.shape), x.dtype)
        N, C = x.shape[0], x.shape[-1]
        vox = x.shape[1] * x.shape[2] * x.shape[3]
        mr = torch.empty((N, C, 2), device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        st = _stream()
        ctx.relu = bool(relu)
        if vox <= CNORM_SMALL_VOX:               # low-resolution stages / semantic maps: one launch
            _l.check(_L().rsuper_cnorm_small(_ptr(x), None, None, _ptr(y), _ptr(mr), N, vox, C, int(relu), float(eps), 0, st), 'cnorm_small')
            ctx.save_for_backward(x, mr)
            return y
        rows = _L().rsuper_cnorm_rows(vox)
        part = torch.empty((N, rows, C, 2), device=x.device, dtype=torch.float32)
        _l.check(_L().rsuper_cnorm_forward(_ptr(x), _ptr(part), _ptr(mr), _ptr(y), N, vox, C, int(relu), float(eps), st), 'cnorm_forward')
        ctx.save_for_backward(x, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mr = ctx.saved_tensors
        dy = dy.contiguous()
        N, C = x.shape[0], x.shape[-1]
        vox = x.shape[1] * x.shape[2] * x.shape[3]
        if vox <= CNORM_SMALL_VOX:
            dx = torch.empty_like(x)
            _l.check(_L().rsuper_cnorm_small(_ptr(x), _ptr(dy), _ptr(mr), _ptr(dx), None, N, vox, C, int(ctx.relu), 0.0, 1, _stream()), 'cnorm_small_bwd')
            return dx, None, None
        rows = _L().rsuper_cnorm_rows(vox)
        part = torch.empty((N, rows, C, 2), device=x.device, dtype=torch.float32)
        gm = torch.empty((N, C, 2), device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        st = _stream()
        _l.check(_L().rsuper_cnorm_backward(_ptr(x), _ptr(dy), _ptr(mr), _ptr(part), _ptr(gm), _ptr(dx), N, vox, C, int(ctx.relu), st), 'cnorm_backward')
        return dx, None, None


class DepthwiseConvFn(torch.autograd.Function):
    """Conv3d(C, C, 3, padding=1, groups=C, bias=False) on a channels-last fp32 tensor (N, D, H, W, C) -- the depthwise member of
    MedFormer's DepthwiseSeparableConv / MBConv (model/dim3/conv_layers.py:126-157, :198-240).  csrc/depthwise.hip."""

    @staticmethod
    def forward(ctx, x, w):
        if not x.is_cuda:
            raise _l.RSuperHipError('DepthwiseConvFn needs a device tensor (no CPU fallback)')
        assert x.dim() == 5 and x.is_contiguous() and x.dtype == torch.float32 and x.shape[-1] % 4 == 0, (tuple(x.shape), x.dtype)
        N, D, H, W, C = x.shape
        assert tuple(w.shape) == (C, 1, 3, 3, 3) and w.dtype == torch.float32
        wc = w.contiguous()
        y = torch.empty_like(x)
        _l.check(_L().rsuper_depthwise3_fwd(_ptr(x), _ptr(wc), _ptr(y), N, D, H, W, C, 0, _stream()), 'depthwise3_fwd')
        ctx.save_for_backward(x, wc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, D, H, W, C = x.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _l.check(_L().rsuper_depthwise3_fwd(_ptr(dy), _ptr(w), _ptr(dx), N, D, H, W, C, 1, _stream()), 'depthwise3_bwd_data')
        if ctx.needs_input_grad[1]:
            rows = _L().rsuper_depthwise3_rows(N * D * H * W)
            part = torch.empty((rows, 27, C), device=x.device, dtype=torch.float32)
            dw = torch.empty_like(w)
            _l.check(_L().rsuper_depthwise3_wgrad(_ptr(x), _ptr(dy), _ptr(part), _ptr(dw), N, D, H, W, C, _stream()), 'depthwise3_wgrad')
        return dx, dw


def ball_search(x, diameter, sigma):
    """Gaussian-ball correlation of a non-negative (D, H, W) f32 map with the ball kernel of odd `diameter` + first arg-max
    (training/losses_foundation.py:1435-1446: F.conv3d + torch.argmax): returns an int64[1] key on the device, value bits in the upper and
    0xFFFFFFFF - linear index in the lower half.  Separable two-stage form from diameter 5 (k^2 gathers per voxel instead of k^3 taps)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
    D, H, W = x.shape
    best = torch.zeros(1, device=x.device, dtype=torch.int64)
    ws = (torch.empty((_L().rsuper_ball_workspace_floats(D, H, W, int(diameter)),), device=x.device, dtype=torch.float32)
          if (diameter >= 5 and os.environ.get('RSUPER_BALL_DIRECT', '0') != '1') else None)
    _l.check(_L().rsuper_ball_conv_argmax(_ptr(x), D, H, W, int(diameter), float(sigma), _ptr(best), None, _ptr(ws), _stream()), 'ball_conv_argmax')
    return best


def dilate_volume(vol_u8, kernel_size):
    """dilate_volume (training/losses_foundation.py:22-46) on a uint8 0/1 tensor (..., D, H, W)."""
    return dilate_volume_flags(vol_u8, kernel_size)[0]


def dilate_volume_flags(vol_u8, kernel_size, flags=None):
    """dilate_volume plus the per-volume any-flags of the INPUT (uint8, 0 = that volume and hence its dilation is all zero; None when not computed).
    flags given (e.g. PackedBits.class_flags(): taken from the packed bytes): used as they are -- volumes flagged 0 are not read at all, so they need not
    even have been written (PackedBits.planes(..., with_flagged=True))."""
    v = vol_u8.contiguous()
    assert v.dtype == torch.uint8 and v.dim() >= 3
    D, H, W = v.shape[-3:]
    nvol = v.numel() // (D * H * W)
    out = torch.empty_like(v)
    ks = kernel_size + 1 if kernel_size % 2 == 0 else kernel_size
    tmp = torch.empty_like(v) if ks > 7 else None
    if flags is not None:
        assert flags.numel() == nvol and flags.dtype == torch.uint8
    elif nvol > 1 and (D * H * W) % 16 == 0 and v.data_ptr() % 16 == 0:
        # one HBM-rate pass marks the volumes that are entirely zero (most label planes have no unknown / segment voxels);
        # the dilation passes write zeros for those without reading them
        flags = torch.empty(nvol, device=v.device, dtype=torch.uint8)
        _l.check(_L().rsuper_plane_any(_ptr(v), nvol, D * H * W, _ptr(flags), _stream()), 'plane_any')
    _l.check(_L().rsuper_dilate_volume_sparse(_ptr(v), _ptr(out), _ptr(tmp), _ptr(flags), nvol, D, H, W, kernel_size, _stream()),
             'dilate_volume')
    return out, flags


# ------------------------------------------------------------------------------------------------ pointwise GEMM
_PW_SEEN = {}      # (data_ptr, mode) -> weakref of a weight pointwise_gemm had to pack by itself: candidates of the next pointwise_prepack
_PW_PACKED = {}    # (data_ptr, mode, dtype code) -> (validity key, arena, byte offset)
_PW_TABLES = {}    # dtype code -> (device table, total items, arena, entries)
_PW_SEEN_DIRTY = False
_PW_CAPTURED = []  # (table, arena) pairs a hipGraph capture recorded pointers to: never freed (see pointwise_prepack)


def _pw_key(w):
    return (WEIGHTS_EPOCH, w._version)              # a view shares its base's version counter


def pointwise_prepack(compute):
    """The MFMA fragments of every weight the pointwise GEMMs packed on their own so far (forward and data-gradient layouts), in ONE launch
    -- a module calls it at the top of its forward; later pointwise_gemm calls of this step find their fragments ready and launch only the GEMM
    (MedFormer: 156 pack launches per step -> 1).  Entries are valid for one (WEIGHTS_EPOCH, parameter version): the fused optimiser's update
    invalidates them.  A changed set of weights rebuilds the device table -- never inside a hipGraph capture (a host copy), where the calls
    then pack individually as before."""
    global _PW_SEEN_DIRTY
    dt = _DT[compute]
    tab = _PW_TABLES.get(dt)
    if tab is None or _PW_SEEN_DIRTY:
        if not _PW_SEEN or torch.cuda.is_current_stream_capturing():
            return
        live = []
        for (ptr, mode), (ref, shape) in list(_PW_SEEN.items()):
            base = ref()                                        # the parameter (the GEMM may have seen a 2-D view of a Conv3d weight)
            if base is None or base.data_ptr() != ptr or base.numel() != shape[0] * shape[1]:
                del _PW_SEEN[(ptr, mode)]
            else:
                live.append((ptr, mode, shape, ref))
        if not live:
            return
        KS = 8 if compute == torch.float32 else 16
        rows, off, item, entries = [], 0, 0, []
        for ptr, mode, (R_, C_), ref in live:
            N, K = (R_, C_) if mode == 0 else (C_, R_)
            ntiles, ksteps = (N + 31) // 32, (K + KS - 1) // KS
            rows.append([ptr, off, R_, C_, mode, ntiles, ksteps, item])
            entries.append(((ptr, mode, dt), off, ref))
            off += ntiles * ksteps * 64 * 16
            item += ntiles * ksteps * 64
        rows.append([0, 0, 0, 0, 0, 1, 1, item])
        dev = live[0][3]().device
        tab = (torch.tensor(rows, dtype=torch.int64).to(dev), item, torch.empty((off,), device=dev, dtype=torch.uint8), entries)
        for d in list(_PW_TABLES):                               # one table per compute type, all from the same set of weights
            del _PW_TABLES[d]
        _PW_PACKED.clear()                                       # entries of the old arena (they would keep it alive)
        _PW_TABLES[dt] = tab
        _PW_SEEN_DIRTY = False
    table, total, arena, entries = tab
    for key, off, ref in entries:                                # a parameter that is gone invalidates the table (its memory may be anybody's by now)
        if ref() is None:
            del _PW_TABLES[dt]
            _PW_PACKED.clear()                                   # ... and every fragment set packed from it: a new parameter at a recycled address with
            _PW_SEEN_DIRTY = True                                # the same (epoch, version) would otherwise multiply with the dead one's fragments
            return
    if torch.cuda.is_current_stream_capturing() and not any(t is table for t, _ in _PW_CAPTURED):
        # the graph being captured keeps raw pointers to this table and arena (this launch and every GEMM that reads its fragments from the
        # arena); the cache above drops them when another set of weights shows up (an EMA copy, a second model) or a parameter dies --
        # a later replay would then read a freed descriptor table and write fragments into somebody else's memory.  Captured tables stay
        # alive for the life of the process (a few MB per captured model).
        _PW_CAPTURED.append((table, arena))
    _l.check(_L().rsuper_pointwise_pack_batch(dt, _ptr(table), len(entries), total, _ptr(arena), _stream()), 'pointwise_pack_batch')
    ep = WEIGHTS_EPOCH
    for key, off, ref in entries:
        _PW_PACKED[key] = ((ep, ref()._version), arena, off, ref)


def pointwise_gemm(x2, w, bias, mode, compute, res=None):
    """csrc/pointwise.hip on f32 rows.  mode 0: y = x2 @ w.T (+ bias), w (N, K); mode 1: y = x2 @ w, w (K, N) (the data gradient of mode 0).
    compute: torch.bfloat16 (bf16 MFMA, fp32 accumulate) or torch.float32 (exact-f32 MFMA); storage stays fp32.
    res (R, N) f32 contiguous: added after the bias in the epilogue (identity shortcuts)."""
    import weakref
    # x2 may be a column slice of a wider row-major matrix (stride(1) == 1, row stride a multiple of 4): the kernel takes the leading dimension
    assert x2.dim() == 2 and x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.stride(0) % 4 == 0 and w.dtype == torch.float32 and w.is_contiguous()
    R, K = x2.shape
    N = w.shape[0] if mode == 0 else w.shape[1]
    assert (w.shape[1] if mode == 0 else w.shape[0]) == K
    dt = _DT[compute]
    y = torch.empty((R, N), device=x2.device, dtype=torch.float32)
    hit = _PW_PACKED.get((w.data_ptr(), mode, dt))
    base = w._base if w._base is not None else w
    if hit is not None and hit[0] == _pw_key(w) and hit[3]() is base:     # fragments from this step's pointwise_prepack, of THIS parameter (not of a
        wptr, packed = None, hit[1].data_ptr() + hit[2]                   # dead one whose address and version counter it inherited)
    else:
        if isinstance(base, torch.nn.Parameter) and base.data_ptr() == w.data_ptr() and base.numel() == w.numel():
            k = (w.data_ptr(), mode)
            if k not in _PW_SEEN or _PW_SEEN[k][0]() is not base:
                global _PW_SEEN_DIRTY
                _PW_SEEN[k] = (weakref.ref(base), tuple(w.shape))
                _PW_SEEN_DIRTY = True
        ws = torch.empty((_L().rsuper_pointwise_packed_bytes(dt, K, N),), device=x2.device, dtype=torch.uint8)
        wptr, packed = _ptr(w), _ptr(ws)
    assert res is None or (res.shape == (R, N) and res.dtype == torch.float32 and res.is_contiguous())
    _l.check(_L().rsuper_pointwise(dt, mode, _ptr(x2), x2.stride(0), wptr, _ptr(bias) if bias is not None else None, _ptr(res), N, _ptr(y), N, R, K, N, packed,
                                   _stream()), 'pointwise')
    return y


def pointwise_wgrad(dy2, x2, want_bias, compute):
    """dW = dy2^T x2 (N, K) and, if wanted, db = dy2.sum(0): csrc/pointwise.hip, slabs of the rows added in slab order (deterministic)."""
    assert dy2.dim() == 2 and x2.dim() == 2 and dy2.shape[0] == x2.shape[0] and dy2.dtype == x2.dtype == torch.float32
    assert dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 4 == 0 and x2.stride(0) % 4 == 0      # column slices of wider matrices are fine
    R, N = dy2.shape
    K = x2.shape[1]
    S = _L().rsuper_pointwise_wgrad_splits(R, N, K)
    ws = torch.empty((S * (N * K + N),), device=dy2.device, dtype=torch.float32)
    dw = torch.empty((N, K), device=dy2.device, dtype=torch.float32)
    db = torch.empty((N,), device=dy2.device, dtype=torch.float32) if want_bias else None
    _l.check(_L().rsuper_pointwise_wgrad(_DT[compute], _ptr(dy2), dy2.stride(0), _ptr(x2), x2.stride(0), R, N, K, _ptr(ws), S, _ptr(dw), _ptr(db) if want_bias else None,
                                         _stream()), 'pointwise_wgrad')
    return dw, db


def pointwise_supported(x, w):
    """Shapes the HIP pointwise GEMM takes (f32 rows on the GPU, channel counts multiples of 4, below 4 GiB)."""
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0
            and (x.numel() // x.shape[-1] + 1) * max(w.shape[0], w.shape[1]) * 4 < (1 << 32))


# ------------------------------------------------------------------------------------------------ dispatcher registration
# every *Fn above becomes the torch.library op rsuper::<name> (schema + CUDA kernel + autograd formula); `XFn.apply` keeps working
from . import library as _library   # noqa: E402

if os.environ.get('RSUPER_NO_TORCH_LIBRARY', '0') != '1':     # =1: the plain autograd.Function classes (A/B of the dispatcher's host cost)
    _library.install()
