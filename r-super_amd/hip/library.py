"""torch.library registration of the HIP ops: every operator of the hot path is a dispatcher op `rsuper::<name>` with a schema, a CUDA
(= HIP on ROCm) kernel registration and an autograd formula (an AutogradCUDA registration, see below) -- "the conv / norm / loss ops are
registered as custom HIP ops behind the repo's existing model/ and training/ interfaces" (BASELINE.json north_star).  The modules call
`torch.ops.rsuper.*` through the `*Fn.apply` names they always used (hip/ops.py rebinds them to the registered ops at import).

    torch.ops.rsuper.basic_block(xa, mra, xb, mrb, w1, w2, ws, pack_bufs, pack_bns, stride) -> (out, mr_out)
    torch.ops.rsuper.maxpool2(x) -> (y, mr)                    torch.ops.rsuper.upsample_trilinear(x, size) -> (y, mr)
    torch.ops.rsuper.maxpool2_skip(x) -> (y, mr, x)            (x again as the skip connection: both gradients of x meet in one backward launch)
    torch.ops.rsuper.stem_conv(img, w, dtype) -> (y, mr)       torch.ops.rsuper.head_conv(x, w, b) -> logits
    torch.ops.rsuper.conv3(x, w) -> y                          torch.ops.rsuper.channel_norm(x, eps, relu) -> y
    torch.ops.rsuper.depthwise_conv3(x, w) -> y                torch.ops.rsuper.squeeze_excite(x, w1, b1, w2, b2) -> y
    torch.ops.rsuper.bidir_attention(fqv, mqv, heads, scale) -> (f_out, m_out)      torch.ops.rsuper.cl_planar(x, K) -> planes
loss ops (training/losses_foundation.py; reference call sites rsuper_train/training/losses_foundation.py:945-956, 541-607, 22-99, 1387-1532):
    torch.ops.rsuper.plane_partials(logits, x_off, xstride, planes, kinv, t, k, w1, w2, kflags, tpk, tpk_classes) -> (sums of term 0, sums of the other terms)
    torch.ops.rsuper.seg_from_sums(sums, cw, B, C, V, scale) -> loss
    torch.ops.rsuper.dilate_volume(vol, kernel_size) -> vol      torch.ops.rsuper.ball_search(x, diameter, sigma) -> key   (no derivative: masks / indices)

Only a "CUDA" kernel is registered: a CPU tensor reaches no kernel and raises (the product path has no CPU fallback).

How the registration is written: the kernels of an op and its derivative already exist as the `forward` / `backward` pair of a
torch.autograd.Function in hip/ops.py (ctypes launches on the current HIP stream).  `_register` defines the schema and registers two
kernels: at the CUDA key the forward launches alone (what runs under inference_mode / below autograd), at the AutogradCUDA key the
Function itself -- forward launches plus the autograd node whose backward launches the derivative kernels.  (torch.library's
`register_autograd` wrapper was tried first: in torch 2.10 it leaves ~7 cyclic Python objects per call, closures over the argument
tensors, for the cycle collector -- 357 per UNet step -- and a collection that frees device tensors inside a hipGraph
capture aborts the process.)  The derivative launches raw kernels: eager autograd and hipGraph capture are the execution models here,
not a tracing compiler.
"""
import torch

from . import ops as _ops

LIB = torch.library.Library('rsuper', 'DEF')


class _Ctx:
    """The ctx surface an autograd.Function-style forward touches, for the forward-only CUDA kernel (nothing is kept)."""
    needs_input_grad = ()

    def save_for_backward(self, *ts):
        pass

    def mark_non_differentiable(self, *ts):
        pass

    def set_materialize_grads(self, v):
        pass


def _register_plain(name, schema, fn):
    """Define rsuper::<name> for an operator without a derivative (uint8 masks, indices): one kernel at the CUDA key."""
    LIB.define(name + schema)
    LIB.impl(name, fn, 'CUDA')
    return getattr(torch.ops.rsuper, name)


def _register(name, schema, Fn, to_fn_args=None):
    """Define rsuper::<name> with `schema`; CUDA kernel = Fn.forward alone, AutogradCUDA kernel = Fn.apply (forward + autograd node).
    to_fn_args: schema arguments -> Fn.forward arguments (default: identical)."""
    LIB.define(name + schema)

    def kernel(*args):
        c = _Ctx()
        c.needs_input_grad = (False,) * 16
        return Fn.forward(c, *(to_fn_args(*args) if to_fn_args else args))

    def with_autograd(*args):
        return Fn.apply(*(to_fn_args(*args) if to_fn_args else args))

    LIB.impl(name, kernel, 'CUDA')
    LIB.impl(name, with_autograd, 'AutogradCUDA')
    return getattr(torch.ops.rsuper, name)


class _Apply:
    """`XFn.apply(...)` as the modules call it, routed to the registered dispatcher op."""

    def __init__(self, op, adapt=None, doc=None):
        self.op, self.adapt, self.__doc__ = op, adapt, doc

    def apply(self, *args):
        return self.op(*(self.adapt(*args) if self.adapt else args))


def _bb_fn_args(xa, mra, xb, mrb, w1, w2, ws, pack_bufs, pack_bns, stride):
    return xa, mra, xb, mrb, w1, w2, ws, (None if pack_bufs is None else (list(pack_bufs), list(pack_bns))), stride


def _bb_apply_args(xa, mra, xb, mrb, w1, w2, ws, packs=None, stride=1):
    return xa, mra, xb, mrb, w1, w2, ws, (None if packs is None else list(packs[0])), (None if packs is None else list(packs[1])), stride


def install():
    """Register every op and rebind the `*Fn` names of hip/ops.py to the dispatcher entries (idempotent)."""
    if getattr(_ops, '_LIBRARY_INSTALLED', False):
        return
    reg = [
        ('BasicBlockFn', 'basic_block',
         '(Tensor xa, Tensor mra, Tensor? xb, Tensor? mrb, Tensor w1, Tensor w2, Tensor? ws, Tensor[]? pack_bufs, int[]? pack_bns, int stride) -> (Tensor, Tensor)',
         _bb_fn_args, _bb_apply_args),
        ('MaxPoolFn', 'maxpool2', '(Tensor x) -> (Tensor, Tensor)', None, None),
        ('MaxPoolSkipFn', 'maxpool2_skip', '(Tensor(a) x) -> (Tensor, Tensor, Tensor(a))', None, None),
        ('UpsampleFn', 'upsample_trilinear', '(Tensor x, int[] size) -> (Tensor, Tensor)', lambda x, size: (x, tuple(size)), None),
        ('StemFn', 'stem_conv', '(Tensor img, Tensor w, ScalarType dtype) -> (Tensor, Tensor)', None, None),
        ('HeadFn', 'head_conv', '(Tensor x, Tensor w, Tensor b) -> Tensor', None, None),
        ('Conv3Fn', 'conv3', '(Tensor x, Tensor w) -> Tensor', None, None),
        ('PlanarFn', 'cl_planar', '(Tensor x, int K) -> Tensor', None, None),
        ('SqueezeExciteFn', 'squeeze_excite', '(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2) -> Tensor', None, None),
        ('BidirAttnFn', 'bidir_attention', '(Tensor fqv, Tensor mqv, int heads, float scale) -> (Tensor, Tensor)', None, None),
        ('TokenAttnFn', 'token_attention', '(Tensor qkv, int heads, float scale) -> Tensor', None, None),
        ('ChannelNormFn', 'channel_norm', '(Tensor x, float eps, bool relu) -> Tensor', None, None),
        ('DepthwiseConvFn', 'depthwise_conv3', '(Tensor x, Tensor w) -> Tensor', None, None),
    ]
    for cls, name, schema, to_fn, adapt in reg:
        Fn = getattr(_ops, cls)
        op = _register(name, schema, Fn, to_fn)
        setattr(_ops, '_' + cls, Fn)                              # the kernel pair itself stays reachable (tests, tools)
        setattr(_ops, cls, _Apply(op, adapt, Fn.__doc__))
    _ops._LIBRARY_INSTALLED = True


def install_loss_ops(lf):
    """Register the loss operators of training/losses_foundation.py (called at the end of that module; idempotent): the fused plane sums and
    the segmentation loss from them with their derivatives, the ball dilation and the ball search as plain CUDA kernels."""
    if getattr(lf, '_LIBRARY_INSTALLED', False):
        return

    def pp_to_fn(logits, x_off, xstride, planes, kinv, t, k, w1, w2, kflags, tpk, tpk_classes):
        from ..training.dataset.packed import PackedBits
        return logits, [lf._Term(x_off[i], xstride[i], planes[i], t[i], k[i], w1[i], w2[i], kinv[i], kflags[i],
                                 None if tpk[i] is None else PackedBits(tpk[i], tpk_classes[i])) for i in range(len(planes))]

    def pp_adapt(logits, terms):
        return (logits, [tm.x_off for tm in terms], [tm.xstride for tm in terms], [tm.planes for tm in terms], [tm.kinv for tm in terms],
                [tm.t for tm in terms], [tm.k for tm in terms], [tm.w1 for tm in terms], [tm.w2 for tm in terms], [tm.kflags for tm in terms],
                [None if tm.tpk is None else tm.tpk.packed for tm in terms], [0 if tm.tpk is None else tm.tpk.C for tm in terms])

    pp = _register('plane_partials', '(Tensor logits, int[] x_off, int[] xstride, int[] planes, bool[] kinv, Tensor?[] t, Tensor?[] k, '
                   'Tensor?[] w1, Tensor?[] w2, Tensor?[] kflags, Tensor?[] tpk, int[] tpk_classes) -> (Tensor, Tensor)', lf._PartialsFn, pp_to_fn)
    sf = _register('seg_from_sums', '(Tensor sums, Tensor? cw, int B, int C, int V, float scale) -> Tensor', lf._SegFromSums)
    dv = _register_plain('dilate_volume', '(Tensor vol, int kernel_size) -> Tensor', _ops.dilate_volume)
    bs = _register_plain('ball_search', '(Tensor x, int diameter, float sigma) -> Tensor', _ops.ball_search)
    lf._PartialsFnImpl, lf._SegFromSumsImpl = lf._PartialsFn, lf._SegFromSums
    lf._PartialsFn = _Apply(pp, pp_adapt, lf._PartialsFnImpl.__doc__)
    lf._SegFromSums = _Apply(sf, None, lf._SegFromSumsImpl.__doc__)
    _ops._dilate_volume_impl, _ops._ball_search_impl = _ops.dilate_volume, _ops.ball_search
    _ops.dilate_volume, _ops.ball_search = dv, bs
    lf._LIBRARY_INSTALLED = True
