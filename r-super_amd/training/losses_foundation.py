"""R-Super losses on gfx950 kernels, behind the interface of rsuper_train/training/losses_foundation.py
(line numbers below refer to that file).

    calculate_loss(model_output, label, unk_voxels, args, matcher, chosen_segment_mask, tumor_volumes_report,
                   tumor_diameters, classes, input_tensor=None, class_weights=None, ...) -> dict with 'overall'

Structure: every per-voxel pass (masked BCE, Dice partial sums, soft volumes, ball-loss weighted BCE, ball
dilation, ball correlation + argmax, top-k selection, rank weights) is a HIP kernel called through the C ABI;
one fused autograd node per network head produces all partial sums and, in backward, writes d(logits) in one
pass.  The tiny (B,C) algebra (adaptive Tversky alpha :581-586 -- NOT detached, clamps, means, weights) stays
in torch autograd so gradients flow exactly as in the reference.
Host-synchronous control flow of ball_loss (argmax -> centre, growth / dilation loops, :1444-1461,:1513-1522)
is kept host-synchronous, as in the reference.
"""
import math
import os

import numpy as np
import torch

from ..hip import lib as _l
from ..hip import ops
from ..hip.ops import _ptr, _stream, _L

SANITY_CHECKS = os.environ.get('RSUPER_SANITY', '1') != '0'   # reference-style input checks / NaN guard (host syncs)
# A train_ddp.StepGuard: the same guards without host synchronisation -- the violated condition is a device flag the training loop reads one step
# later and raises from with the reference's message (the optimiser skips the update of a step whose gradient norm is not finite, so the weights are
# those of the last good step when it does).  None: the checks synchronise where the reference does.
GUARD = None


# ------------------------------------------------------------------------------------------------ helpers
def _u8(t):
    if t.dtype == torch.uint8:
        return t.contiguous()
    return (t != 0).to(torch.uint8).contiguous()


def _is_packed(v):
    from .dataset.packed import PackedBits
    return isinstance(v, PackedBits)


def _lesion_planes(v, chs):
    """uint8 (B, C, D, H, W) view of a label / unknown / segment volume for the REPORT losses, which index the lesion channels `chs` only (:286-297,
    :1571-1605): a tensor is taken as it is; of a bit-packed volume (dataset.packed.PackedBits) only those planes are inflated -- the other
    planes of the returned tensor are never written and must not be read."""
    return v.planes(chs) if _is_packed(v) else _u8(v)


def _sample_any(v, as_bool=True):
    """(B,) any() per sample of a uint8 tensor or a bit-packed volume."""
    return v.sample_any(as_bool) if _is_packed(v) else _plane_any(v, 1, as_bool=as_bool)


def lesion_channel_lists(classes):
    """get_lesion_channels (:204-221): lesion/cyst/pdac/pnet channels grouped per organ, in class order; every group is
    the list of channels the reference max-merges (:218-219)."""
    groups = {}
    for i, c in enumerate(classes):
        for suffix in ['lesion', 'cyst', 'pdac', 'pnet']:
            if suffix in c:
                name = c[:c.index('_' + suffix) + len('_' + suffix)].replace('pancreatic', 'pancreas')
                groups.setdefault(name, []).append(i)
    return groups


def lesion_groups(classes):
    """Single-channel view {group name: channel} used by the fused loss kernels; a group spanning several channels has
    to be merged first (`merge_lesion_channels`), which calculate_loss does on its own."""
    groups = lesion_channel_lists(classes)
    for k, v in groups.items():
        if len(set(v)) != 1:
            raise NotImplementedError(f'lesion group {k!r} spans several channels {v}: merge them with merge_lesion_channels first')
    return {k: v[0] for k, v in groups.items()}


def merge_lesion_channels(t, classes):
    """get_lesion_channels (:204-228): (B, C, ...) -> (B, L, ...) with every organ's lesion sub-channels max-merged
    (logits: max, differentiable through the arg-max channel; 0/1 masks: OR).  Returns (merged, group names)."""
    groups = lesion_channel_lists(classes)
    cols = []
    for idx in groups.values():
        m = t[:, idx[0]]
        for i in idx[1:]:
            m = torch.maximum(m, t[:, i])
        cols.append(m)
    return torch.stack(cols, dim=1).contiguous(), list(groups.keys())


def dilate_volume(volume, kernel_size, full_pass_radius=3):
    """dilate_volume (:22-46); accepts float/bool/uint8 0-1 tensors, returns the input dtype."""
    assert full_pass_radius == 3
    out = ops.dilate_volume(_u8(volume), int(kernel_size))
    return out if volume.dtype == torch.uint8 else out.to(volume.dtype)


def get_known_voxels(y, unk_voxels, dilation=5, sanity=False, classes=None):
    """(:150-165) 1 - dilate(unk, dilation), float like the reference."""
    u = _u8(unk_voxels)
    if dilation > 0:
        u = ops.dilate_volume(u, dilation)
    return (1 - u).to(torch.float32)


def dice_based_volume_loss(x, y, tolerance=0.1, E=500, cross_entropy=False):
    """(:352-395) on (B,L) predicted / report volumes."""
    loss = torch.abs(x - y) / (x + y + E)
    v = torch.max((1 - tolerance) * y, y.clamp(max=100))
    loss = loss - torch.abs(v - y) / (v + y + E)
    loss = torch.clamp(loss, min=0, max=1)
    if cross_entropy:
        loss = -torch.log(torch.ones_like(loss) - loss + 1e-5)
    return loss


# ------------------------------------------------------------------------------------------------ fused partial sums
class _Term:
    """One family of planes of the logits tensor: plane p lives at element offset x_off + p * xstride."""
    __slots__ = ('x_off', 'xstride', 'planes', 't', 'k', 'w1', 'w2', 'kinv', 'kflags', 'tpk')

    def __init__(self, x_off, xstride, planes, t=None, k=None, w1=None, w2=None, kinv=False, kflags=None, tpk=None):
        self.x_off, self.xstride, self.planes, self.t, self.k, self.w1, self.w2 = x_off, xstride, planes, t, k, w1, w2
        self.kinv = bool(kinv)       # k is the dilated UNKNOWN mask: voxel weight = 1 - k (no `1 - dilate(unk)` tensor)
        self.kflags = kflags         # with kinv: uint8 [planes], 0 = the plane of k is all zero (not read)
        self.tpk = tpk               # dataset.packed.PackedBits: the target in bit-packed form (replaces t)

    def targs(self):
        """(t, tpk, tP, tC, k, kflags) pointers / sizes for rsuper_plane_partials_fwd2 / _bwd2."""
        kf = _ptr(self.kflags) if (self.kinv and self.kflags is not None) else None
        if self.tpk is not None:
            return (None, _ptr(self.tpk.packed), int(self.tpk.packed.shape[1]), int(self.tpk.C), _ptr(self.k), kf)
        return (_ptr(self.t), None, 0, 0, _ptr(self.k), kf)


class _PartialsFn(torch.autograd.Function):
    """sums[planes, 6] = (S, A, B, Cn, F1, F2) per term (csrc/loss.hip); backward writes d(logits) once:
    term 0 must cover every plane (segmentation term), later terms accumulate on their planes.
    Returns (sums of term 0, sums of all later terms stacked along the plane axis or an empty (0, 6) tensor): one accumulator buffer and
    one f64 -> f32 conversion per call."""

    @staticmethod
    def forward(ctx, logits, terms):
        assert logits.is_contiguous() and logits.dtype == torch.float32
        V = logits[0, 0].numel()
        total = sum(tm.planes for tm in terms)
        # per-block sums stored, then added in block order (rsuper_plane_partials_fwd3 / rsuper_plane_sums_reduce): no float atomic, no zero fill, no
        # f64 -> f32 conversion pass
        nb = _L().rsuper_plane_partials_blocks(V)
        pblk = torch.empty((total, nb, 6), device=logits.device, dtype=torch.float64)
        row = 0
        for tm in terms:
            _l.check(_L().rsuper_plane_partials_fwd3(_ptr(logits, tm.x_off), tm.xstride, *tm.targs(), _ptr(tm.w1), _ptr(tm.w2),
                                                     _ptr(pblk, row * nb * 6), 2 if tm.kinv else 0, tm.planes, V, _stream()), 'plane_partials_fwd')
            row += tm.planes
        out = torch.empty((total, 6), device=logits.device, dtype=torch.float32)
        _l.check(_L().rsuper_plane_sums_reduce(_ptr(pblk), total, nb, _ptr(out), _stream()), 'plane_sums_reduce')
        ctx.terms = terms
        ctx.save_for_backward(logits)
        return out[:terms[0].planes], out[terms[0].planes:]

    @staticmethod
    def backward(ctx, g0, grest):
        (logits,) = ctx.saved_tensors
        V = logits[0, 0].numel()
        B, C = logits.shape[:2]
        assert ctx.terms[0].planes == B * C and ctx.terms[0].x_off == 0
        dl = torch.empty_like(logits)
        grest = None if grest is None else grest.contiguous().float()
        row = 0
        for i, tm in enumerate(ctx.terms):
            g = g0 if i == 0 else (None if grest is None else grest[row:row + tm.planes])
            if i > 0:
                row += tm.planes
            if g is None:
                if i == 0:
                    dl.zero_()
                continue
            g = g.contiguous().float()
            _l.check(_L().rsuper_plane_partials_bwd2(_ptr(logits, tm.x_off), tm.xstride, *tm.targs(), _ptr(tm.w1), _ptr(tm.w2),
                                                     _ptr(g), _ptr(dl, tm.x_off), (0 if i == 0 else 1) | (2 if tm.kinv else 0), tm.planes, V,
                                                     _stream()),
                      'plane_partials_bwd')
        return dl, None


class _SegFromSums(torch.autograd.Function):
    """scale * (masked BCE mean + DiceLossMultiClass) from the (B*C, 6) sums of the label planes, one launch forward (which also
    writes the Jacobian) -- replaces ~60 launch-bound ATen kernels of (B, C) algebra per step (csrc/loss.hip seg_from_sums_kernel)."""

    @staticmethod
    def forward(ctx, sums, cw, B, C, V, scale):
        sums = sums.contiguous()
        assert sums.dtype == torch.float32 and sums.shape == (B * C, 6)
        if cw is not None:
            cw = cw.contiguous().float()
        loss = torch.empty((), device=sums.device, dtype=torch.float32)
        d = torch.empty_like(sums)
        _l.check(_L().rsuper_seg_from_sums(_ptr(sums), _ptr(cw), B, C, V, float(scale), _ptr(loss), _ptr(d), _stream()), 'seg_from_sums')
        ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None, None, None, None, None


class _ReportFromSums(torch.autograd.Function):
    """(ball_loss_bce, ball_loss_dice, dice_volume_loss) from the (R, 6) sums of the report terms, one launch that also writes the three Jacobians
    (csrc/loss.hip report_from_sums_kernel).  The same algebra used to run on the host through torch's CPU autograd (the sums came over in one copy,
    the gradient went back in one): two pipeline drains and ~0.6 ms of idle GPU per config-3 step; as ATen device ops it is ~200 launch-bound kernels."""

    @staticmethod
    def forward(ctx, rest, roww, plan, flags, rvol, B, L, V, use_vol, tol, nplans, apply_dice, standard_ce):
        rest = rest.contiguous()
        R = rest.shape[0]
        loss = torch.empty(3, device=rest.device, dtype=torch.float32)
        jac = torch.empty((3, R, 6), device=rest.device, dtype=torch.float32)
        _l.check(_L().rsuper_report_from_sums(_ptr(rest), _ptr(roww), R, B, L, V, int(use_vol), _ptr(flags), _ptr(rvol), float(tol), 500.0, nplans,
                                              _ptr(plan), int(apply_dice), int(standard_ce), _ptr(loss), _ptr(jac), _stream()), 'report_from_sums')
        ctx.save_for_backward(jac)
        return loss[0], loss[1], loss[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        (jac,) = ctx.saved_tensors
        g = torch.stack([g0, g1, g2]).view(3, 1, 1)
        return ((jac * g).sum(0),) + (None,) * 12



def _dice_from_sums(A, Bs, Cn, w=None):
    """DiceLossMultiClass (:541-607) from per-(n,c) sums A=sum P, Bs=sum P*T, Cn=sum T (P,T already masked)."""
    TP, FP, FN = Bs, A - Bs, Cn - Bs
    alpha = (FP.sum(0) / (FP.sum(0) + FN.sum(0) + 1e-5)).unsqueeze(0).expand_as(TP).clamp(0.2, 0.8)
    dice = TP / (TP + alpha * FP + (1 - alpha) * FN + 1e-5)
    loss = 1 - dice
    if w is not None:
        loss = loss * w
    return loss.mean()


# ------------------------------------------------------------------------------------------------ ball machinery (host-synchronous control)
def _odd_ceil(v):
    c = math.ceil(v)
    return c + 1 if c % 2 == 0 else c


def ball_kernel_geometry(diameter):
    """create_ball_kernel (:1192-1207): (odd diameter, kernel edge)."""
    d_odd = _odd_ceil(diameter)
    return d_odd, _odd_ceil(1.2 * d_odd)


def ball_nnz(d_odd):
    """Voxels of the ball 4*|o|^2 <= d_odd^2 (support of create_ball_kernel)."""
    r = d_odd // 2
    a = np.arange(-r, r + 1)
    d2 = a[:, None, None] ** 2 + a[None, :, None] ** 2 + a[None, None, :] ** 2
    return int((4 * d2 <= d_odd * d_odd).sum())


def _insert_ball(shape, center, diameter, margin, device):
    """insert_ball (:1336-1385) -> (uint8 mask, voxel count)."""
    d_odd, ks = ball_kernel_geometry(diameter * (1 + margin))
    D, H, W = shape
    out = torch.empty(shape, device=device, dtype=torch.uint8)
    cnt = torch.zeros(1, device=device, dtype=torch.int32)
    _l.check(_L().rsuper_insert_ball(_ptr(out), D, H, W, int(center[0]), int(center[1]), int(center[2]), d_odd, ks // 2, _ptr(cnt), _stream()),
              'insert_ball')
    return out, int(cnt.item())


def _topk_mask(x, ball, k):
    """Exact top-k of x*ball (x >= 0) as a uint8 mask; ties -> lower linear index (radix select on the f32 bits)."""
    V = x.numel()
    k = int(min(k, V))
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    if k <= 0:
        return out.zero_()
    if os.environ.get('RSUPER_TOPK_HOST', '0') != '1':
        # device-resident radix select: one C call, no device->host reads
        ws = torch.empty(260, device=x.device, dtype=torch.int32)
        _l.check(_L().rsuper_topk_select(_ptr(x), _ptr(ball), V, k, _ptr(out), _ptr(ws), _stream()), 'topk_select')
        return out
    prefix, remaining, ties = 0, k, 0
    hist = torch.empty(256, device=x.device, dtype=torch.int32)
    for shift in (24, 16, 8, 0):
        hist.zero_()
        _l.check(_L().rsuper_radix_hist(_ptr(x), _ptr(ball), V, prefix, shift, _ptr(hist), _stream()), 'radix_hist')
        h = hist.cpu().numpy().astype(np.int64)
        acc = 0
        for digit in range(255, -1, -1):       # largest values first
            if acc + h[digit] >= remaining:
                prefix |= digit << shift
                remaining -= acc
                ties = int(h[digit])          # after the last pass: number of elements equal to the threshold
                break
            acc += h[digit]
    # `prefix` = bit pattern of the k-th largest value; `remaining` = how many elements equal to it are still needed
    need = 0xFFFFFFFF if remaining >= ties else remaining      # all ties wanted -> order-free parallel marking
    _l.check(_L().rsuper_topk_mark(_ptr(x), _ptr(ball), V, prefix, need, _ptr(out), _stream()), 'topk_mark')
    return out


def _isolate_params(diameter, tumor_volume, V, volume_margin):
    """The data-independent part of isolate_tumor (:1400-1433, :1466-1481): odd diameter, target volume (raised to the ball's own voxel count - 1
    when the report's volume is smaller; the products then run in float32 as in the reference) and the three top-k sizes (tumour, small, big)."""
    diameter = int(np.round(diameter).astype(int))
    vol = int(np.round(tumor_volume).astype(int))
    if diameter % 2 == 0:
        diameter += 1
    nnz = ball_nnz(diameter)
    f32 = False
    if nnz > vol:                         # :1431-1433 (vol becomes a 0-dim tensor there -> float32 products below)
        vol, f32 = nnz - 1, True
    t = min(V - 1, vol)
    ms = min(0.5, volume_margin)
    t_small = int(np.float32(t) * np.float32(1 - ms)) if f32 else int(t * (1 - ms))
    t_small = max(t_small, min(100, vol))
    t_big = min(V - 1, int(np.float32(vol) * np.float32(1 + volume_margin)) if f32 else int(vol * (1 + volume_margin)))
    return diameter, vol, (t, t_small, t_big)


def isolate_tumor(x, diameter, gaussian, gaussian_std, tumor_volume, diameter_margin=0.5, volume_margin=0.5):
    """(:1387-1532) x: (D,H,W) f32 >= 0 on device.  Returns three uint8 masks (tumor, small, big)."""
    assert gaussian, 'the reference always calls isolate_tumor with gaussian=True (:1713)'
    D, H, W = x.shape
    V = x.numel()
    diameter, vol, ks = _isolate_params(diameter, tumor_volume, V, volume_margin)
    best = ops.ball_search(x, diameter, float(gaussian_std * (diameter / 2.0)))      # rsuper::ball_search (separable two-stage correlation + arg-max)
    key = int(best.item()) & 0xFFFFFFFFFFFFFFFF
    idx = 0xFFFFFFFF - (key & 0xFFFFFFFF)
    center = np.unravel_index(idx, (D, H, W))
    ball, bsum = _insert_ball((D, H, W), center, diameter, diameter_margin, x.device)
    new_dim = diameter
    while bsum < vol:                     # :1450-1461
        old = new_dim
        new_dim = int(np.round(new_dim * 1.1))
        if old == new_dim:
            new_dim += 1
        if new_dim % 2 == 0:
            new_dim += 1
        if new_dim >= max(D, H, W):
            break
        ball, bsum = _insert_ball((D, H, W), center, new_dim, diameter_margin, x.device)
    if min(ks) > 0 and os.environ.get('RSUPER_TOPK_HOST', '0') != '1':
        # the three selections share x and the ball: one batched radix select, AND with the ball folded into the marking (:1504-1507)
        import ctypes
        out3 = torch.empty((3,) + tuple(x.shape), device=x.device, dtype=torch.uint8)
        ws3 = torch.empty(3 * 260, device=x.device, dtype=torch.int32)
        _l.check(_L().rsuper_topk_select_multi(_ptr(x), _ptr(ball), V, (ctypes.c_uint * 3)(*ks), 3, _ptr(out3), _ptr(ws3), 1, _stream()),
                 'topk_select_multi')
        masks = [out3[0], out3[1], out3[2]]
    else:
        masks = [_topk_mask(x, ball, k) for k in ks]
        for m in masks:                       # "ensure no tumor_mask value is outside the ball" (:1504-1507)
            _l.check(_L().rsuper_mask_op(_ptr(m), _ptr(ball), V, 0, _stream()), 'mask_and')
    iters = 0
    while vol < 50 ** 3 and _count(masks[0]) < vol * 0.7:     # :1513-1522
        if iters > 5:
            break
        nm = []
        for m in masks:
            dm = ops.dilate_volume(m, 7)
            _l.check(_L().rsuper_mask_op(_ptr(dm), _ptr(ball), V, 0, _stream()), 'mask_and')
            nm.append(dm)
        masks = nm
        iters += 1
    return masks[0], masks[1], masks[2]


def isolate_tumor_spec(x, diameter, gaussian_std, tumor_volume, checks, diameter_margin=0.5, volume_margin=0.5):
    """isolate_tumor without device->host reads, valid under two assumptions that hold for almost every tumour: the first ball already holds
    `vol` voxels (no growth loop, :1450-1461) and the top-k mask keeps >= 70 % of `vol` inside the ball (no dilation rounds, :1513-1522).  The
    two counts that decide this stay on the device; `checks` collects (ball count, mask count, vol) and the caller reads them all in ONE copy
    after the last tumour of the sample -- if any assumption fails it repeats the sample with isolate_tumor proper.  Returns the masks or None
    when the preconditions of the device-resident selection do not hold (then the caller falls back at once)."""
    D, H, W = x.shape
    V = x.numel()
    diameter, vol, ks = _isolate_params(diameter, tumor_volume, V, volume_margin)
    if min(ks) <= 0 or os.environ.get('RSUPER_TOPK_HOST', '0') == '1':
        return None
    import ctypes
    best = ops.ball_search(x, diameter, float(gaussian_std * (diameter / 2.0)))
    d_odd, kedge = ball_kernel_geometry(diameter * (1 + diameter_margin))
    ball = torch.empty((D, H, W), device=x.device, dtype=torch.uint8)
    cnts = torch.zeros(2, device=x.device, dtype=torch.int32)          # [ball voxels, voxels of the top-k mask inside the ball]
    _l.check(_L().rsuper_insert_ball_at(_ptr(ball), D, H, W, _ptr(best), d_odd, kedge // 2, _ptr(cnts), _stream()), 'insert_ball_at')
    out3 = torch.empty((3, D, H, W), device=x.device, dtype=torch.uint8)
    ws3 = torch.empty(3 * 260, device=x.device, dtype=torch.int32)
    _l.check(_L().rsuper_topk_select_multi(_ptr(x), _ptr(ball), V, (ctypes.c_uint * 3)(*ks), 3, _ptr(out3), _ptr(ws3), 1, _stream()), 'topk_select_multi')
    _l.check(_L().rsuper_count(_ptr(out3), V, _ptr(cnts, 1), _stream()), 'count')
    checks.append((cnts, vol))
    return out3[0], out3[1], out3[2]


def _spec_ok(checks, extra=None):
    """ONE device->host copy for all tumours of a sample (+ `extra`, a device int32 tensor the caller wants along): True when no tumour needed the
    growth loop or the dilation rounds.  Returns (ok, extra values)."""
    parts = [c for c, _ in checks] + ([extra] if extra is not None else [])
    h = torch.cat(parts).cpu().numpy()
    ok = True
    for i, (_, vol) in enumerate(checks):
        bsum, c0 = int(h[2 * i]), int(h[2 * i + 1])
        if bsum < vol or (vol < 50 ** 3 and c0 < vol * 0.7):
            ok = False
    return ok, (h[2 * len(checks):] if extra is not None else None)


SPECULATIVE_BALL_SEARCH = os.environ.get('RSUPER_BALL_SPEC', '1') == '1'
GWRP_SORT_ABOVE = int(os.environ.get('RSUPER_GWRP_SORT_ABOVE', '16000'))     # pseudo masks above this size rank by sorting (32 k voxels: 690 -> 228 us; below: one O(n^2) launch wins)


def _plane_any(t, lead_dims, as_bool=True):
    """any() over the trailing volume of a contiguous uint8 tensor, one flag per leading index: (B, ...) -> bool tensor of
    shape t.shape[:lead_dims] on the device (HIP kernel at HBM rate instead of an ATen byte reduction)."""
    t = t if t.is_contiguous() else t.contiguous()
    shape = t.shape[:lead_dims]
    planes = int(np.prod(shape)) if len(shape) else 1
    V = t.numel() // max(planes, 1)
    flags = torch.empty(planes, device=t.device, dtype=torch.uint8)
    if (V % 16 == 0 or planes == 1) and t.data_ptr() % 16 == 0:
        _l.check(_L().rsuper_plane_any(_ptr(t), planes, V, _ptr(flags), _stream()), 'plane_any')
        return flags.view(shape).bool() if as_bool else flags.view(shape)
    r = t.flatten(lead_dims).any(lead_dims)
    return r if as_bool else r.to(torch.uint8)


def _count(m):
    c = torch.zeros(1, device=m.device, dtype=torch.int32)
    _l.check(_L().rsuper_count(_ptr(m), m.numel(), _ptr(c), _stream()), 'count')
    return int(c.item())


def gwrp_foreground_weights(x_plane, pm, c=0.5, N=None):
    """GlobalWeightedRankPooling(sig(x)*pm + pm, N=|pm|, c, return_weights=True, hard_cutoff=True) * |pm| * pm
    (:1780-1791): rank the pseudo-mask voxels by sig(x) (desc, ties by index), weight d^rank, renormalise."""
    V = x_plane.numel()
    if N is None:
        N = _count(pm)
    w = torch.zeros(x_plane.shape, device=x_plane.device, dtype=torch.float32)
    if N == 0:
        return w, 0
    sig = torch.empty(x_plane.shape, device=x_plane.device, dtype=torch.float32)
    _l.check(_L().rsuper_sigmoid_mask(_ptr(x_plane), _ptr(pm), _ptr(sig), V, _stream()), 'sigmoid_mask')
    vals = torch.empty(N, device=w.device, dtype=torch.float32)
    idx = torch.empty(N, device=w.device, dtype=torch.int32)
    n = torch.zeros(1, device=w.device, dtype=torch.int32)
    _l.check(_L().rsuper_compact(_ptr(sig), _ptr(pm), V, _ptr(vals), _ptr(idx), _ptr(n), _stream()), 'compact')
    d = float(np.float32(1 - c) ** (np.float32(1.0) / np.float32(max(N, 1))))      # d = (1-c)^(1/N)  (:482)
    s_n = (1.0 - d ** N) / (1.0 - d) if d < 1.0 else float(N)                       # sum_{r<N} d^r
    if N > GWRP_SORT_ABOVE:
        # large pseudo masks: ranks from two sorts (voxel index ascending, then value descending, stable: ties by lower index -- the order the
        # pairwise-count kernel defines) and the same weight formula, O(n log n) instead of O(n^2) (33 k voxels: milliseconds -> ~0.1 ms)
        ids, perm = torch.sort(idx[:N].long())
        order = torch.sort(vals[:N][perm], descending=True, stable=True).indices
        ranked = ids[order].contiguous()
        _l.check(_L().rsuper_rank_assign(_ptr(ranked), N, math.log2(d), float(N / s_n), _ptr(w), _stream()), 'rank_assign')
        return w, N
    _l.check(_L().rsuper_rank_weights(_ptr(vals), _ptr(idx), N, math.log2(d), float(N / s_n), _ptr(w), _stream()), 'rank_weights')
    return w, N


class _BallPlan:
    """Masks for one batch item of ball_loss, built without gradient from the current logits."""
    __slots__ = ('kind', 'b', 'c', 'pm', 'penal', 'fw', 'big', 'pens', 'chs')


def _pre_key(label_u8, unk_u8, mask_u8, volumes, diameters, chs):
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (label_u8, unk_u8, mask_u8, volumes, diameters)) + (tuple(chs),)


def _ball_inputs(label_u8, unk_u8, mask_u8, volumes, diameters, chs, prefetch):
    """The part of _ball_plans that reads only the batch: dilated segment masks, the penalised region, and the host copies of the report
    volumes / diameters / per-channel segment flags.  prefetch: the three host copies are asynchronous (pinned memory + an event)."""
    m_l = mask_u8[:, chs].contiguous()
    u_l = unk_u8[:, chs].contiguous()
    t_l = label_u8[:, chs].contiguous()
    mseg = ops.dilate_volume(m_l, 31)                                   # :1593
    # to_penalize = ((1 - unk)*(1 - labels) + segment) > 0   (:1597-1605); unk dilation 1 is the identity
    pen = (((1 - u_l) * (1 - t_l)) + mseg > 0).to(torch.uint8)
    srcs = (volumes.detach().float(), diameters.detach().float(), _plane_any(mseg, 2))
    if prefetch:
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in srcs]
        for h, t in zip(host, srcs):
            h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    else:
        host, ev = [t.cpu() for t in srcs], None
    return dict(key=_pre_key(label_u8, unk_u8, mask_u8, volumes, diameters, chs), mseg=mseg, pen=pen, vols=host[0], dias=host[1], seg_any=host[2],
                event=ev, keep=srcs)


def _volume_flags(label_u8, mseg31, tumor_volumes_report, chs):
    """(B, 2L) flags [tumour annotated per voxel | segment present] and the (B, 1) report volume of the volume loss (:313, :335)."""
    flags = torch.stack([_plane_any(label_u8[:, c], 1) for c in chs] + [_plane_any(m, 1) for m in mseg31], 1).float()
    return flags, tumor_volumes_report.float().sum(-1, keepdim=True)


def _unknown_planes(unk_voxels, chs):
    """(uint8 planes, per-plane any-flags or None) of the unknown-voxel map: a tensor as it is (flags computed by the dilation); of a bit-packed map the
    planes that hold a voxel at all plus the lesion planes, with the flags read from the packed bytes."""
    if _is_packed(unk_voxels):
        return unk_voxels.planes(chs, with_flagged=True), unk_voxels.class_flags()
    return _u8(unk_voxels), None


def prepare_report_supervision(label, unk_voxels, chosen_segment_mask, tumor_volumes_report, tumor_diameters, classes, args):
    """Call BEFORE the forward pass of a training step (train_ddp.train_step does): the batch-only inputs of the ball loss -- two dilations and
    three small host reads -- are queued ahead of the network, so that calculate_loss(pre=...) does not start with a blocking copy.  Without
    it the first host read drains the launch queue right after the forward, and the ~100 launches of the ball search that follow run at the
    host's launch rate with the GPU idle in between (config 3: the report losses cost 3.0 ms per step, 1.1 ms of it kernel time).
    Returns None when the step has no ball loss to prepare (or for lesion groups spanning several channels, which merge their tensors first)."""
    rw = float(getattr(args, 'report_volume_loss_basic', 0.0))
    if rw <= 0 or not ('ball' in args.loss or 'dynamic' in args.loss or 'dll' in args.loss) or chosen_segment_mask is None:
        return None
    if tumor_volumes_report is None or tumor_diameters is None or not label.is_cuda:
        return None
    if any(len(v) > 1 for v in lesion_channel_lists(classes).values()):
        return None
    chs = list(lesion_groups(classes).values())
    if not chs:
        return None
    label_u8 = _lesion_planes(label, chs)
    unk_u8 = _unknown_planes(unk_voxels, chs)[0] if unk_voxels is not None else torch.zeros(tuple(label.shape), device=label.device, dtype=torch.uint8)
    mask_u8 = _lesion_planes(chosen_segment_mask, chs)
    with torch.no_grad():
        pre = _ball_inputs(label_u8, unk_u8, mask_u8, tumor_volumes_report, tumor_diameters, chs, True)
        if 'both' in args.loss:                                             # the volume loss runs too: its dilated masks and (B, 2L + 1) flags
            pre['mseg31'] = [ops.dilate_volume(mask_u8[:, c].contiguous(), 31) for c in chs]
            pre['flags'], pre['rvol'] = _volume_flags(label_u8, pre['mseg31'], tumor_volumes_report, chs)
    pre['u8'] = (label, unk_voxels, chosen_segment_mask, label_u8, unk_u8, mask_u8)     # calculate_loss reuses the uint8 views (same key)
    return pre


def _ball_plans(out, label_u8, unk_u8, mask_u8, volumes, diameters, groups, margin, pre=None):
    """The data-dependent, non-differentiable part of ball_loss (:1587-1737): returns one plan per sample.
    pre: what prepare_report_supervision computed from the same batch before the forward pass (else it is computed here)."""
    B, C, D, H, W = out.shape
    chs = list(groups.values())
    L = len(chs)
    V = D * H * W
    if pre is not None and ('pen' not in pre or pre['key'] != _pre_key(label_u8, unk_u8, mask_u8, volumes, diameters, chs)):
        pre = None                                                      # other tensors, or already consumed by another head (pen is edited in place below)
    if pre is None:
        pre = _ball_inputs(label_u8, unk_u8, mask_u8, volumes, diameters, chs, False)
    mseg, pen = pre['mseg'], pre.pop('pen')
    if pre['event'] is not None:
        pre['event'].synchronize()                                      # recorded before the forward was queued: long done, no pipeline drain
    vols_h, dias_h, seg_any = pre['vols'].numpy(), pre['dias'].numpy(), pre['seg_any'].numpy()
    plans = []
    for b in range(B):
        p = _BallPlan()
        p.b, p.chs = b, chs
        if SANITY_CHECKS:
            assert np.array_equal(dias_h[b].sum(-1) > 0, vols_h[b] > 0), 'Tumor diameters and volumes should be consistent'
            assert seg_any[b].sum() <= 1, 'Only one channel should be non-zero'
        if not seg_any[b].any() or vols_h[b].sum() == 0:               # :1625
            p.kind, p.pens = 'none', pen[b].contiguous()
            plans.append(p)
            continue
        li = int(np.nonzero(seg_any[b])[0][0])
        c = chs[li]
        penal = pen[b, li].contiguous()
        tumor_seg = mseg[b].sum(0).clamp(max=1).to(torch.uint8).contiguous()       # one channel active -> 0/1
        order = [int(i) for i in np.argsort(-vols_h[b], kind='stable') if vols_h[b][int(i)] > 0]
        def tumour_loop(spec):
            x_it = torch.empty((D, H, W), device=out.device, dtype=torch.float32)
            _l.check(_L().rsuper_sigmoid_mask(_ptr(out, (b * C + c) * V), _ptr(tumor_seg), _ptr(x_it), V, _stream()), 'sigmoid_mask')
            pm_small, pm_big, checks = None, None, []
            for ti in order:                                            # :1695-1719
                vol, dmax = float(vols_h[b][ti]), float(dias_h[b][ti].max())
                if dmax <= 1:
                    dmax = 3
                if vol <= 1:
                    vol = 9
                if spec:
                    r3 = isolate_tumor_spec(x_it, dmax, 1.5, vol, checks, margin, margin)
                    if r3 is None:
                        return None
                    pm, pms, pmb = r3
                else:
                    pm, pms, pmb = isolate_tumor(x_it, dmax, True, 1.5, vol, margin, margin)
                _l.check(_L().rsuper_zero_where(_ptr(x_it), _ptr(pm), V, _stream()), 'zero_where')    # x_iter *= (1 - pseudo_mask)
                if pm_small is None:
                    pm_small, pm_big = pms, pmb
                else:
                    _l.check(_L().rsuper_mask_op(_ptr(pm_small), _ptr(pms), V, 1, _stream()), 'mask_or')
                    _l.check(_L().rsuper_mask_op(_ptr(pm_big), _ptr(pmb), V, 1, _stream()), 'mask_or')
            return pm_small, pm_big, checks

        npm_known = None
        res3 = tumour_loop(True) if SPECULATIVE_BALL_SEARCH else None
        if res3 is not None:
            # the counts that validate the speculation and |pseudo mask| (needed on the host by the rank weights) come over in one copy
            npm_dev = torch.zeros(1, device=out.device, dtype=torch.int32)
            _l.check(_L().rsuper_count(_ptr(res3[0]), V, _ptr(npm_dev), _stream()), 'count')
            ok, extra = _spec_ok(res3[2], npm_dev)
            if ok:
                npm_known = int(extra[0])
            else:
                res3 = None
        if res3 is None:
            res3 = tumour_loop(False)
        pm_small, pm_big = res3[0], res3[1]
        big = ops.dilate_volume(pm_big, 7)                              # :1727-1731
        # border = (BIG - PM) > 0 ; penalize *= (1 - border)  ==  penal &= ~(BIG & ~PM)
        border = big.clone()
        _l.check(_L().rsuper_mask_op(_ptr(border), _ptr(pm_small), V, 2, _stream()), 'mask_andnot')
        _l.check(_L().rsuper_mask_op(_ptr(penal), _ptr(border), V, 2, _stream()), 'mask_andnot')
        fw, npm = gwrp_foreground_weights(out[b, c], pm_small, N=npm_known)
        if SANITY_CHECKS:
            assert npm > 0, 'Pseudo mask should have at least one voxel'
        p.kind, p.c, p.pm, p.penal, p.fw, p.big = 'tumor', c, pm_small, penal, fw, big
        plans.append(p)
    return plans


# ------------------------------------------------------------------------------------------------ calculate_loss
def _calculate_loss_merged(model_output, label, unk_voxels, args, matcher, chosen_segment_mask, tumor_volumes_report,
                           tumor_diameters, classes, input_tensor, class_weights):
    """Lesion groups spanning several channels (e.g. pancreatic_lesion_1 / _2).  The reference max-merges the sub-channels of
    every tensor the report losses read (get_lesion_channels, :204-221, called at :286-297 and :1571-1583) while the
    segmentation term keeps all C channels (:945-957).  Same arithmetic here as two passes of the fused single-channel path:
      1. segmentation term on the original C channels (report losses off);
      2. report terms on the merged problem -- L channels named after the groups, logits max-merged (the gradient flows to
         the arg-max sub-channel, as torch.max does in the reference), labels / unknown / segment masks OR-merged, class
         weights max-merged -- with the segmentation weight set to zero."""
    import copy
    rw = float(args.report_volume_loss_basic)
    a_seg = copy.copy(args)
    a_seg.report_volume_loss_basic = 0.0
    seg = calculate_loss(model_output, label, unk_voxels, a_seg, matcher, chosen_segment_mask, tumor_volumes_report, tumor_diameters,
                         _delesioned(classes), input_tensor, class_weights)
    loss = {'segmentation': seg['segmentation']}
    if rw > 0:
        result = model_output['segmentation']
        deep = isinstance(result, (tuple, list))
        heads = [merge_lesion_channels(r.float(), classes)[0] for r in (result if deep else [result])]
        names = list(lesion_channel_lists(classes).keys())
        lab = merge_lesion_channels(_u8(label), classes)[0]
        unk = merge_lesion_channels(_u8(unk_voxels), classes)[0] if unk_voxels is not None else None
        msk = merge_lesion_channels(_u8(chosen_segment_mask), classes)[0] if chosen_segment_mask is not None else None
        cw = None
        if class_weights is not None:
            cw = merge_lesion_channels(class_weights.reshape(class_weights.shape[0], -1).float(), classes)[0]
        a_rep = copy.copy(args)
        a_rep.seg_loss = 0.0
        rep = calculate_loss({'segmentation': heads if deep else heads[0]}, lab, unk, a_rep, matcher, msk, tumor_volumes_report,
                             tumor_diameters, names, input_tensor, cw)
        for k, v in rep.items():
            if k not in ('segmentation', 'overall'):
                loss[k] = v
    else:
        loss['report'] = seg['report']
    overall = None
    for k in list(loss.keys()):
        overall = loss[k] if overall is None else overall + loss[k]
    loss['overall'] = overall
    return loss


def _delesioned(classes):
    """Class names for the segmentation-only pass: the lesion sub-channels keep their positions but lose the suffixes the
    grouping keys on, so no report term (there is none in that pass) and no merge is attempted."""
    out = []
    for c in classes:
        for suffix in ['lesion', 'cyst', 'pdac', 'pnet']:
            c = c.replace(suffix, 'x')
        out.append(c)
    return out


def calculate_loss(model_output, label, unk_voxels, args, matcher, chosen_segment_mask,
                   tumor_volumes_report, tumor_diameters, classes, input_tensor=None, class_weights=None,
                   model_genesis=False, clip_only=False, report_embeddings=None, dist=None, pre=None):
    """Same contract as the reference (:685-1076): returns {'segmentation', report keys..., 'overall'}."""
    if model_genesis or clip_only or getattr(args, 'classification_branch', False) or getattr(args, 'multi_ch_tumor', False):
        raise NotImplementedError('model_genesis / clip_only / classification_branch / multi_ch_tumor are baselines outside '
                                  'the accelerated R-Super path (SURVEY.md section 2.1)')
    _l.require_device()
    merged = any(len(v) > 1 for v in lesion_channel_lists(classes).values())
    label_pk = None
    if merged:      # lesion groups spanning several channels merge whole tensors first (_calculate_loss_merged): inflate
        label, unk_voxels, chosen_segment_mask = (v.unpack() if _is_packed(v) else v for v in (label, unk_voxels, chosen_segment_mask))
    elif _is_packed(label):
        # bit-packed label: the segmentation term reads the bits (rsuper_plane_partials_fwd2 / _bwd2); the report terms index the lesion channels only, of which
        # _lesion_planes inflates the planes -- with or without report supervision nothing else of the volume is ever inflated
        label_pk = label
    if merged:
        return _calculate_loss_merged(model_output, label, unk_voxels, args, matcher, chosen_segment_mask, tumor_volumes_report,
                                      tumor_diameters, classes, input_tensor, class_weights)
    result = model_output['segmentation']
    deep = isinstance(result, (tuple, list))
    heads = list(result) if deep else [result]
    B, C = label.shape[:2]
    assert len(classes) == C, f'Number of classes in classes: {len(classes)} does not match the number of channels in label: {C}'
    assert len(classes) == heads[0].shape[1], 'Number of classes in result does not match the number of channels in label'
    if pre is not None and pre['u8'][0] is label and pre['u8'][1] is unk_voxels and pre['u8'][2] is chosen_segment_mask:
        label_u8, unk_u8, mask_u8 = pre['u8'][3:]                        # the tensors prepare_report_supervision keyed its results on
    else:
        pre = None
        chs_ = list(lesion_groups(classes).values())
        report_on = float(args.report_volume_loss_basic) > 0
        label_u8 = _u8(label) if label_pk is None else (_lesion_planes(label, chs_) if report_on else None)
        zeros = lambda: torch.zeros(tuple(label.shape), device=label.device, dtype=torch.uint8)
        unk_u8 = _unknown_planes(unk_voxels, chs_)[0] if unk_voxels is not None else zeros()
        mask_u8 = (_lesion_planes(chosen_segment_mask, chs_) if (report_on or not _is_packed(chosen_segment_mask)) else None) \
            if chosen_segment_mask is not None else zeros()
    D, H, W = label.shape[2:]
    V = D * H * W

    # (the per-sample any() of a bit-packed volume comes from its packed bytes: the uint8 views above hold the lesion planes only)
    m_src = chosen_segment_mask if _is_packed(chosen_segment_mask) else mask_u8
    u_src = unk_voxels if _is_packed(unk_voxels) else unk_u8
    if SANITY_CHECKS and chosen_segment_mask is not None and GUARD is not None:      # :864-869 without the host round trips
        GUARD.consistency(_sample_any(m_src, as_bool=False), _sample_any(u_src, as_bool=False), tumor_volumes_report)
    elif SANITY_CHECKS and chosen_segment_mask is not None:                # :864-869
        m_any = _sample_any(m_src).cpu()
        if bool(m_any.any()):
            u_any = _sample_any(u_src).cpu()
            v_any = (tumor_volumes_report.sum(1) != 0).cpu()
            for b in range(B):
                if m_any[b] and not u_any[b]:
                    raise ValueError('unk_voxels should not be all zeros if chosen_segment_mask is not all zeros')
                if m_any[b] and not v_any[b]:
                    raise ValueError('tumor_volumes_report should not be all zeros if chosen_segment_mask is not all zeros')

    if class_weights is not None and torch.equal(class_weights, torch.ones_like(class_weights)):     # :876-877
        class_weights = None
    cw = None
    if class_weights is not None:
        cw = class_weights.to(label.device).float()
        assert cw.shape == (B, C), f'Class weights should be (B, C), got {tuple(cw.shape)}'

    # :899 / get_known_voxels :150; known = 1 - unk5.  unk_any[b * C + c] = 0: that plane of the map has no unknown voxel at all (nor has its dilation)
    unk5, unk_any = ops.dilate_volume_flags(unk_u8, 5, flags=unk_voxels.class_flags() if _is_packed(unk_voxels) else None) \
        if unk_voxels is not None else (None, None)
    groups = lesion_groups(classes)
    chs = list(groups.values())
    L = len(chs)
    rw = float(args.report_volume_loss_basic)

    loss_seg_total = None
    rep = {}
    rep_scalar = None
    mseg31 = None
    for j, r in enumerate(heads):
        r = r.contiguous().float() if (not r.is_contiguous() or r.dtype != torch.float32) else r
        aw = args.aux_weight[j] if deep else 1.0
        use_ball = use_vol = False
        if rw > 0:
            use_ball = ('ball' in args.loss or 'dynamic' in args.loss or 'dll' in args.loss)
            if deep:
                use_ball = use_ball and not (j != 0 and 'last' in args.loss)        # :924
            use_vol = (not use_ball) or ('both' in args.loss)
        terms = [_Term(0, V, B * C, t=None if label_pk is not None else label_u8, k=unk5, kinv=True, kflags=unk_any, tpk=label_pk)]
        if use_vol and L > 0:
            if mseg31 is None:
                mseg31 = pre['mseg31'] if pre is not None and 'mseg31' in pre else \
                    [ops.dilate_volume(mask_u8[:, c].contiguous(), 31) for c in chs]   # :308
            for li, c in enumerate(chs):
                terms.append(_Term(c * V, C * V, B, k=mseg31[li]))
        plans = []
        if use_ball and L > 0:
            with torch.no_grad():
                plans = _ball_plans(r.detach(), label_u8, unk_u8, mask_u8, tumor_volumes_report, tumor_diameters, groups,
                                    float(args.ball_volume_margin), pre=pre)
            for p in plans:
                if p.kind == 'none':
                    for li, c in enumerate(p.chs):
                        terms.append(_Term((p.b * C + c) * V, 0, 1, k=p.pens[li].contiguous()))
                else:
                    terms.append(_Term((p.b * C + p.c) * V, 0, 1, t=p.pm, k=p.penal, w1=p.fw, w2=p.big))
        seg_sums, rest = _PartialsFn.apply(r, terms)
        # ---- segmentation: masked BCE mean + adaptive-Tversky Dice (:945-956)
        seg = _SegFromSums.apply(seg_sums, cw, B, C, V, aw * args.seg_loss)
        loss_seg_total = seg if loss_seg_total is None else loss_seg_total + seg
        loss_r = {}
        # ---- report terms from their sums (volume loss :250-349, ball loss :1537-1864): one launch, nothing leaves the device
        if rest.shape[0] > 0:
            # class weight of every row = cw[sample, channel] of the term the row belongs to: the (sample, channel) pairs are host knowledge, the gather runs on
            # the device (no device -> host read of the weights: that was a pipeline drain on every head of every step when class weights are given)
            R = rest.shape[0]
            row_b, row_c = [0] * R, [0] * R
            ti = 0
            flags = rvol = None
            if use_vol and L > 0:
                for li, c in enumerate(chs):
                    for b in range(B):
                        row_b[li * B + b], row_c[li * B + b] = b, c
                ti += L * B
                if pre is not None and 'flags' in pre:
                    flags, rvol = pre['flags'], pre['rvol']               # computed before the forward pass (prepare_report_supervision)
                else:
                    flags, rvol = _volume_flags(label_u8, mseg31, tumor_volumes_report, chs)
            plan_l = []
            if use_ball and L > 0:
                for p in plans:
                    if p.kind == 'none':                                  # :1625-1661
                        plan_l += [0, ti]
                        for li, c in enumerate(chs):
                            row_b[ti + li], row_c[ti + li] = p.b, c
                        ti += L
                    else:
                        plan_l += [1, ti]
                        row_b[ti], row_c[ti] = p.b, p.c
                        ti += 1
            assert ti == R
            if cw is None:
                roww_d = torch.ones(R, dtype=torch.float32, device=rest.device)
            else:
                bc = torch.tensor([row_b, row_c], dtype=torch.int64).to(rest.device, non_blocking=True)
                roww_d = cw.detach().float()[bc[0], bc[1]].contiguous()
            plan_d = torch.tensor(plan_l, dtype=torch.int32).to(rest.device, non_blocking=True) if plan_l else None
            lb, ld, lv = _ReportFromSums.apply(rest, roww_d, plan_d, None if flags is None else flags.contiguous().float(),
                                               None if rvol is None else rvol.contiguous().float().view(-1), B, max(L, 1), V, use_vol and L > 0,
                                               float(args.volume_loss_tolerance), len(plan_l) // 2, 'dice' in args.loss,
                                               bool(getattr(args, 'stardard_ce_ball', False)))
            if use_vol and L > 0:
                loss_r['dice_volume_loss'] = lv
            if use_ball and L > 0:
                loss_r['ball_loss_bce'] = lb
                loss_r['ball_loss_dice'] = ld if 'dice' in args.loss else torch.zeros_like(lb)
        if not loss_r and rep_scalar is None:
            rep_scalar = torch.zeros((), device=r.device)                   # aw * rw * 0 for every head
        for k, v in loss_r.items():
            wk = {'ball_loss_bce': args.ball_bce_weight, 'ball_loss_dice': args.ball_dice_weight}.get(k, 1)
            term = aw * rw * wk * v
            rep[k] = rep[k] + term if k in rep else term

    loss = {'segmentation': loss_seg_total}
    if rep:
        # key order of the reference: ball keys first, then volume
        for k in ('ball_loss_bce', 'ball_loss_dice', 'dice_volume_loss'):
            if k in rep:
                loss[k] = rep[k]
    else:
        loss['report'] = rep_scalar
    overall = None
    for k in list(loss.keys()):
        overall = loss[k] if overall is None else overall + loss[k]
    loss['overall'] = overall
    if SANITY_CHECKS and GUARD is not None:                                # :1070-1071, read one step later (the NaN step's update is skipped on the device)
        GUARD.nan(overall.detach())
    elif SANITY_CHECKS and bool(torch.isnan(overall).any()):               # :1070-1071
        raise ValueError('loss is nan, propagating this can destroy the network weights, STOP!')
    assert overall.requires_grad, 'Loss overall should require grad'
    return loss


# the loss operators as dispatcher ops rsuper::plane_partials / seg_from_sums / dilate_volume / ball_search (hip/library.py)
if os.environ.get('RSUPER_NO_TORCH_LIBRARY', '0') != '1':
    from ..hip import library as _library
    _library.install_loss_ops(__import__(__name__, fromlist=['_']))
