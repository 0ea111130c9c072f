"""GPU parity checks shared by tests/test_gpu_*.py (pytest, -m gpu) and tests/gpu_diag.py (run-everything report).

Every check compares the HIP path (called through the C ABI via rsuper_amd.hip) with the oracle / a plain torch
fp32 CPU reference on the same seeded inputs and returns a dict(name, ok, err, tol, note).
Tolerances: f32 mode 1e-4 (BASELINE.json north_star) relative to the output scale; bf16 mode is judged against the
same fp32 oracle with a stated looser tolerance (the reference is fp32-only).
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import synth  # noqa: E402
from oracle import unet_oracle as uo, losses_oracle as lo, train_oracle as to, morph as omorph  # noqa: E402

DEV = 'cuda'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16}
TOL = {'f32': 1e-4, 'bf16': 3e-2}
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def to_cl(x, dt):
    """(N,C,D,H,W) fp32 cpu -> (N,D,H,W,C) device tensor of dtype dt."""
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV).to(dt)


def from_cl(x):
    return x.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def rnd(x, mode):
    return x.bfloat16().float() if mode == 'bf16' else x.clone()


def relerr(got, ref):
    got, ref = got.double(), ref.double()
    scale = max(ref.abs().max().item(), 1e-12)
    return (got - ref).abs().max().item() / scale


def l2err(got, ref):
    got, ref = got.double().reshape(-1), ref.double().reshape(-1)
    return ((got - ref).norm() / max(ref.norm().item(), 1e-30)).item()


def err_for(mode, got, ref):
    """f32: max-norm relative to the output scale.  bf16: relative L2 -- a handful of ReLU-mask flips caused by the
    bf16 rounding of intermediates makes the max-norm of deep gradients meaningless."""
    return relerr(got, ref) if mode == 'f32' else l2err(got, ref)


def result(name, err, tol, note=''):
    return dict(name=name, ok=bool(err <= tol) and math.isfinite(err), err=float(err), tol=float(tol), note=note)


def stats_ref(x, eps=1e-4):
    """(N,C,D,H,W) -> mean, rstd as (N,C,2) like the kernels' (mean, rstd)."""
    m = x.mean(dim=(2, 3, 4))
    v = ((x - m[:, :, None, None, None]) ** 2).mean(dim=(2, 3, 4))
    return torch.stack([m, 1.0 / torch.sqrt(v + eps)], dim=-1)


def _rng_t(seed, shape, scale=1.0):
    return T(synth.rng(seed).standard_normal(shape).astype(np.float32) * np.float32(scale))


# ================================================================================================ conv kernels
def check_conv_fwd(mode, N, S, Ca, Cb, Cout, fused_sc=False, residual=False, norm=True, seed=0):
    from rsuper_amd.hip import ops
    dt = DT[mode]
    D, H, W = S
    xa = rnd(_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3, mode)
    xb = rnd(_rng_t(seed + 2, (N, Cb, D, H, W)) * 1.5 - 0.2, mode) if Cb else None
    Cin = Ca + Cb
    w1 = _rng_t(seed + 3, (Cout, Cin, 3, 3, 3), 1.0 / math.sqrt(27 * Cin))
    ws = _rng_t(seed + 4, (Cout, Cin, 3, 3, 3), 1.0 / math.sqrt(27 * Cin)) if fused_sc else None
    res = rnd(_rng_t(seed + 5, (N, Cout, D, H, W)), mode) if residual else None
    x = xa if xb is None else torch.cat([xa, xb], 1)
    xh = F.relu(uo.instance_norm(x)) if norm else x
    if mode == 'bf16':
        xh = xh.bfloat16().float()
    wr1 = rnd(w1, mode)
    ref = F.conv3d(xh, wr1, padding=1)
    if fused_sc:
        ref = torch.cat([ref, F.conv3d(xh, rnd(ws, mode), padding=1)], 1)
    if residual:
        ref = ref + res
    # device
    mra = stats_ref(xa).to(DEV) if norm else None
    mrb = stats_ref(xb).to(DEV) if (norm and xb is not None) else None
    nc = Cout * (2 if fused_sc else 1)
    bn = ops.pick_bn(nc, dt, dims=(N, D, H, W), epi=0)
    wp = ops.pack_weights(dt, 0, w1.to(DEV), ws.to(DEV) if ws is not None else None, Ca, Cb, Cout, Cout if fused_sc else 0, bn)
    out = torch.empty((N, D, H, W, nc), device=DEV, dtype=dt)
    part = ops.part_buffer(dt, (N, D, H, W), nc, bn, DEV, fill=float('nan'))
    a = ops.Src(to_cl(xa, dt), mr=mra)
    b = ops.Src(to_cl(xb, dt), mr=mrb) if xb is not None else None
    r = ops.Src(to_cl(res, dt)) if residual else None
    ops.igemm(0, a, b, wp, nc, bn, (N, D, H, W), out, res=r, part=part)
    mr = ops.stats_finalize(part, D * H * W)
    torch.cuda.synchronize()
    got = from_cl(out)
    e1 = relerr(got, ref)
    mr_ref = stats_ref(rnd(ref, mode))
    e2 = relerr(mr.cpu()[..., 0], mr_ref[..., 0]) + relerr(mr.cpu()[..., 1], mr_ref[..., 1])
    tol = TOL[mode]
    return result(f'conv_fwd[{mode} N{N} S{S} {Ca}+{Cb}->{Cout} sc{int(fused_sc)} res{int(residual)} norm{int(norm)}]',
                  max(e1, e2 * (0.1 if mode == 'bf16' else 1.0)), tol, f'out {e1:.2e} stats {e2:.2e}')


def check_conv_mixed_sources(N, S, Ca, Cb, Cout, seed=0):
    """bf16 forward with ONE normalised source (a) and ONE raw source (b) at a shape the default dispatch hands to the depth-reuse kernel when both sources are
    of one kind (>= 400 tiles of 4x8x16, 33..64 columns): that kernel stages both sources the same way and does not take such a launch -- query
    (pick_bn / part_buffer with mixed=True, i.e. src_flags bit 0) and launch must agree on the ordinary kernel instead of failing (api.hip)."""
    from rsuper_amd.hip import ops
    dt = torch.bfloat16
    D, H, W = S
    dims = (N, D, H, W)
    xa = rnd(_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3, 'bf16')
    xb = rnd(torch.relu(_rng_t(seed + 2, (N, Cb, D, H, W)) * 1.5 - 0.2), 'bf16')
    w1 = _rng_t(seed + 3, (Cout, Ca + Cb, 3, 3, 3), 1.0 / math.sqrt(27 * (Ca + Cb)))
    xh = torch.cat([F.relu(uo.instance_norm(xa)).bfloat16().float(), xb], 1)
    ref = F.conv3d(xh, rnd(w1, 'bf16'), padding=1)
    from rsuper_amd.hip import lib as _lib
    assert _lib.lib().rsuper_conv3_kd_bn(ops._DT[dt], 0, N, D, H, W, Cout, 0) == 64, 'pick a shape the depth-reuse kernel takes with uniform sources'
    bn = ops.pick_bn(Cout, dt, dims=dims, epi=0, mixed=True)
    wp = ops.pack_weights(dt, 0, w1.to(DEV), None, Ca, Cb, Cout, 0, bn)
    out = torch.full((N, D, H, W, Cout), float('nan'), device=DEV, dtype=dt)
    part = ops.part_buffer(dt, dims, Cout, bn, DEV, fill=float('nan'), mixed=True)
    ops.igemm(0, ops.Src(to_cl(xa, dt), mr=stats_ref(xa).to(DEV)), ops.Src(to_cl(xb, dt)), wp, Cout, bn, dims, out, part=part)
    mr = ops.stats_finalize(part, D * H * W)
    torch.cuda.synchronize()
    e1 = relerr(from_cl(out), ref)
    mr_ref = stats_ref(rnd(ref, 'bf16'))
    e2 = relerr(mr.cpu()[..., 0], mr_ref[..., 0]) + relerr(mr.cpu()[..., 1], mr_ref[..., 1])
    return result(f'conv_mixed_sources[bf16 N{N} S{S} {Ca}n+{Cb}raw->{Cout}]', max(e1, e2), TOL['bf16'], f'out {e1:.2e} stats {e2:.2e} bn {bn}')


def _block_ref(mode, xa, xb, w1, ws, dy1, dys):
    """Reference for the fused conv1(+shortcut) data/weight gradients on x_hat = relu(IN(x))."""
    x = (xa if xb is None else torch.cat([xa, xb], 1)).clone().requires_grad_(True)
    xh = F.relu(uo.instance_norm(x))
    w1r = rnd(w1, mode).requires_grad_(True)
    wsr = rnd(ws, mode).requires_grad_(True) if ws is not None else None
    y = F.conv3d(xh, w1r, padding=1)
    loss = (y * dy1).sum()
    if ws is not None:
        loss = loss + (F.conv3d(xh, wsr, padding=1) * dys).sum()
    gx, = torch.autograd.grad(loss, x, retain_graph=True)
    gxh, = torch.autograd.grad(loss, xh, retain_graph=True)
    gw = torch.autograd.grad(loss, [w1r] + ([wsr] if ws is not None else []))
    return gx, gxh, gw, xh.detach()


def check_conv_bwd(mode, N, S, Ca, Cb, Cout, fused_sc, seed=0, tr=None):
    """dgrad (epi 1, relu mask + IN sums) + in_bwd_finalize + wgrad against autograd."""
    from rsuper_amd.hip import ops
    dt = DT[mode]
    D, H, W = S
    xa = rnd(_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3, mode)
    xb = rnd(_rng_t(seed + 2, (N, Cb, D, H, W)) * 1.5 - 0.2, mode) if Cb else None
    Cin = Ca + Cb
    w1 = _rng_t(seed + 3, (Cout, Cin, 3, 3, 3), 1.0 / math.sqrt(27 * Cin))
    ws = _rng_t(seed + 4, (Cout, Cin, 3, 3, 3), 1.0 / math.sqrt(27 * Cin)) if fused_sc else None
    dy1 = rnd(_rng_t(seed + 6, (N, Cout, D, H, W)), mode)
    dys = rnd(_rng_t(seed + 7, (N, Cout, D, H, W)), mode) if fused_sc else None
    gx, gxh, gw, xh = _block_ref(mode, xa, xb, w1, ws, dy1, dys)
    # device
    mra = stats_ref(xa).to(DEV)
    mrb = stats_ref(xb).to(DEV) if xb is not None else None
    sa = ops.Src(to_cl(xa, dt), mr=mra)
    sb = ops.Src(to_cl(xb, dt), mr=mrb) if xb is not None else None
    y1 = ops.Src(to_cl(dy1, dt))
    y2 = ops.Src(to_cl(dys, dt)) if fused_sc else None
    bn = ops.pick_bn(Cin, dt, dims=(N, D, H, W), epi=1)
    wp = ops.pack_weights(dt, 1, w1.to(DEV), ws.to(DEV) if ws is not None else None, Cout, Cout if fused_sc else 0, Cin, 0, bn)
    g0 = torch.empty((N, D, H, W, Cin), device=DEV, dtype=dt)
    part = ops.part_buffer(dt, (N, D, H, W), Cin, bn, DEV, fill=float('nan'), epi=1)
    ops.igemm(1, y1, y2, wp, Cin, bn, (N, D, H, W), g0, part=part, ea=sa, eb=sb)
    gm = ops.stats_finalize(part, D * H * W, mode=1)
    if xb is None:
        dxa = ops.in_bwd_finalize(ops.Src(g0), sa, gm, Ca)
        dx = from_cl(dxa)
    else:
        dxa = ops.in_bwd_finalize(ops.Src(g0, C=Ca), sa, gm[:, :Ca].contiguous(), Ca)
        dxb = ops.in_bwd_finalize(ops.Src(g0, C=Cb, off=Ca), sb, gm[:, Ca:].contiguous(), Cb)
        dx = torch.cat([from_cl(dxa), from_cl(dxb)], 1)
    dw1 = torch.zeros((Cout, Cin, 3, 3, 3), device=DEV)
    dws = torch.zeros_like(dw1) if fused_sc else None
    old = os.environ.get('RSUPER_WGRAD_TR')
    if tr is not None:
        os.environ['RSUPER_WGRAD_TR'] = str(tr)
    try:
        ops.wgrad(sa, sb, y1, y2, dw1, dws, (N, D, H, W))
    finally:
        if tr is not None:
            if old is None:
                os.environ.pop('RSUPER_WGRAD_TR')
            else:
                os.environ['RSUPER_WGRAD_TR'] = old
    torch.cuda.synchronize()
    e_g = relerr(from_cl(g0), gxh * (xh > 0))
    e_dx = relerr(dx, gx)
    e_w = relerr(dw1.cpu(), gw[0])
    if fused_sc:
        e_w = max(e_w, relerr(dws.cpu(), gw[1]))
    tol = TOL[mode]
    return result(f'conv_bwd[{mode} tr{tr} N{N} S{S} {Ca}+{Cb}->{Cout} sc{int(fused_sc)}]', max(e_g, e_dx, e_w), tol,
                  f'g {e_g:.2e} dx {e_dx:.2e} dw {e_w:.2e}')


def _xhat_bf16(x, mr):
    """The kernels' prologue on bf16 inputs, restated: x_hat = bf16(max(fma(x, rstd, -mean * rstd), 0)) with f32 constants (csrc/common.hpp norm_relu16);
    the fma is evaluated in float64 and rounded once to f32 (differs from a hardware fma only on exact f32 ties of the f64 value)."""
    sc = mr[:, :, 1].float()
    nb = (-mr[:, :, 0].float() * sc)                                          # f32 product, as the kernel forms it
    v = (x.double() * sc.double()[:, :, None, None, None] + nb.double()[:, :, None, None, None]).float()
    return torch.relu(v).bfloat16().double()


def check_conv_exact(kind, N, S, Ca, Cb, Cout, fused_sc=False, residual=False, norm=True, seed=0):
    """bf16 forward ('fwd') / data gradient ('dgrad') of the igemm kernels against float64 on EXACTLY the bf16 operands the kernel multiplies
    (the forward / dgrad analogue of check_wgrad_xhat).  Per element the bound is one bf16 rounding of the result
    (2^-8 |ref|, round-to-nearest-even of the f32 accumulator) plus the f32 accumulation slack of the K = 27 Cin reduction
    ((K / 8 + 4) roundings of at most 2^-24 sum|a b| each: one per MFMA k-step and lane half); err = max |got - ref| / bound must be <= 1.
    One dropped tap x channel term is ~ sum|a b| / K, five times the slack, on every output it feeds."""
    from rsuper_amd.hip import ops
    dt = torch.bfloat16
    D, H, W = S
    Cin = Ca + Cb
    xa = (_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3).bfloat16().float()
    xb = (_rng_t(seed + 2, (N, Cb, D, H, W)) * 1.5 - 0.2).bfloat16().float() if Cb else None
    w1 = _rng_t(seed + 3, (Cout, Cin, 3, 3, 3), 1.0 / math.sqrt(27 * Cin))
    ws = _rng_t(seed + 4, (Cout, Cin, 3, 3, 3), 1.0 / math.sqrt(27 * Cin)) if fused_sc else None
    wd = [w.bfloat16().double() for w in ((w1, ws) if fused_sc else (w1,))]
    mra = stats_ref(xa)
    mrb = stats_ref(xb) if xb is not None else None
    dims = (N, D, H, W)
    if kind == 'fwd':
        if norm:
            xh = _xhat_bf16(xa, mra) if xb is None else torch.cat([_xhat_bf16(xa, mra), _xhat_bf16(xb, mrb)], 1)
        else:
            xh = (xa if xb is None else torch.cat([xa, xb], 1)).double()
        ref = torch.cat([F.conv3d(xh, w, padding=1) for w in wd], 1)
        mag = torch.cat([F.conv3d(xh.abs(), w.abs(), padding=1) for w in wd], 1)
        res = _rng_t(seed + 5, (N, ref.shape[1], D, H, W)).bfloat16().float() if residual else None
        if residual:
            ref = ref + res.double()
            mag = mag + res.double().abs()
        nc = ref.shape[1]
        bn = ops.pick_bn(nc, dt, dims=dims, epi=0)
        wp = ops.pack_weights(dt, 0, w1.to(DEV), ws.to(DEV) if fused_sc else None, Ca, Cb, Cout, Cout if fused_sc else 0, bn)
        out = torch.full((N, D, H, W, nc), float('nan'), device=DEV, dtype=dt)
        part = ops.part_buffer(dt, dims, nc, bn, DEV, fill=float('nan'))
        a = ops.Src(to_cl(xa, dt), mr=mra.to(DEV) if norm else None)
        b = ops.Src(to_cl(xb, dt), mr=mrb.to(DEV) if norm else None) if xb is not None else None
        ops.igemm(0, a, b, wp, nc, bn, dims, out, res=ops.Src(to_cl(res, dt)) if residual else None, part=part)
        K = 27 * Cin
    else:
        dys = [_rng_t(seed + 6 + i, (N, Cout, D, H, W)).bfloat16().float() for i in range(len(wd))]
        ref = sum(F.conv_transpose3d(dy.double(), w, padding=1) for dy, w in zip(dys, wd))
        mag = sum(F.conv_transpose3d(dy.double().abs(), w.abs(), padding=1) for dy, w in zip(dys, wd))
        x = xa if xb is None else torch.cat([xa, xb], 1)
        mr = mra if xb is None else torch.cat([mra, mrb], 1)
        mask = (x > mr[:, :, 0].float()[:, :, None, None, None]).double()          # sign((x - mean) * rstd) as the epilogue evaluates it in f32
        ref, mag = ref * mask, mag * mask
        nc = Cin
        bn = ops.pick_bn(nc, dt, dims=dims, epi=1)
        wp = ops.pack_weights(dt, 1, w1.to(DEV), ws.to(DEV) if fused_sc else None, Cout, Cout if fused_sc else 0, Cin, 0, bn)
        out = torch.full((N, D, H, W, nc), float('nan'), device=DEV, dtype=dt)
        part = ops.part_buffer(dt, dims, nc, bn, DEV, fill=float('nan'), epi=1)
        sa = ops.Src(to_cl(xa, dt), mr=mra.to(DEV))
        sb = ops.Src(to_cl(xb, dt), mr=mrb.to(DEV)) if xb is not None else None
        ops.igemm(1, ops.Src(to_cl(dys[0], dt)), ops.Src(to_cl(dys[1], dt)) if fused_sc else None, wp, nc, bn, dims, out, part=part, ea=sa, eb=sb)
        K = 27 * Cout * len(wd)
    fin = ops.stats_finalize(part, D * H * W, mode=0 if kind == 'fwd' else 1)
    torch.cuda.synchronize()
    got = from_cl(out).double()
    bound = 2.0 ** -8 * ref.abs() + (K / 8 + 4) * 2.0 ** -24 * mag + 1e-30
    ratio = ((got - ref).abs() / bound)
    ratio = torch.where(torch.isfinite(ratio), ratio, torch.full_like(ratio, float('inf')))
    same = (got == ref.float().bfloat16().double()).double().mean().item()   # elements equal to the correctly rounded float64 result
    # The rows the launch emits for the NEXT kernel (what the one-rounding check did not cover): the statistics the consumer normalises
    # with (forward: mean, rstd) / the InstanceNorm-backward sums (data gradient: mean g, mean g x_n), against float64 sums over the tensor the launch
    # WROTE (the bf16 values a consumer reads).  A kernel may sum its f32 accumulators before rounding: that moves a mean by ~ 2^-9 rms / sqrt(voxels),
    # two orders below the bound used here (1e-4 of the column's rms / of rstd); one dropped or doubled partial row of a few hundred is >= 1e-3.
    fin = fin.cpu().double()
    cnt = D * H * W
    if kind == 'fwd':
        mean = got.mean(dim=(2, 3, 4)); var = (got * got).mean(dim=(2, 3, 4)) - mean * mean
        rms = (got * got).mean(dim=(2, 3, 4)).sqrt().clamp_min(1e-30)
        e_s = max(((fin[..., 0] - mean).abs() / rms).max().item(), (fin[..., 1] * torch.sqrt(var.clamp_min(0) + 1e-4) - 1.0).abs().max().item())
    else:
        xn = (x.double() - mr[:, :, 0].double()[:, :, None, None, None]) * mr[:, :, 1].double()[:, :, None, None, None]
        rms = (got * got).mean(dim=(2, 3, 4)).sqrt().clamp_min(1e-30)
        e_s = max(((fin[..., 0] - got.mean(dim=(2, 3, 4))).abs() / rms).max().item(), ((fin[..., 1] - (got * xn).mean(dim=(2, 3, 4))).abs() / rms).max().item())
    e_s = e_s if math.isfinite(e_s) else float('inf')
    return result(f'conv_exact[{kind} N{N} S{S} {Ca}+{Cb}->{Cout} sc{int(fused_sc)} res{int(residual)} norm{int(norm)}]', max(ratio.max().item(), e_s / 1e-4), 1.0,
                  f'max |got-ref| / (1 bf16 rounding + f32 slack) {ratio.max().item():.3f}; == bf16(float64 result) on {same:.4f} of the elements; '
                  f'emitted rows vs float64 sums of the written tensor {e_s:.1e} (bound 1e-4); bn {bn}')


def check_conv_exact_s2(kind, N, S, Ca, Cout, seed=0, kernel='1'):
    """The stride-2 [conv1 | shortcut] GEMM of BasicBlock(stride=2) (unet_utils.py:38-39) -- persistent strided forward `s2k` / data gradient `s2d` (kernel='1',
    the default) or the parity-class kernel (kernel='0') -- held to the ABSOLUTE bound check_conv_exact uses (round 5 only compared the two
    kernels with each other): float64 F.conv3d(stride=2) / conv_transpose3d(stride=2) on exactly the bf16 operands, one bf16 rounding of the result + f32
    accumulation slack per element, and the emitted statistics / InstanceNorm-backward rows against float64 sums of the written tensor at 1e-4."""
    from rsuper_amd.hip import ops
    dt = torch.bfloat16
    D, H, W = S
    OD, OH, OW = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    dims = (N, D, H, W)
    xa = (_rng_t(seed + 1, (N, Ca, D, H, W)) * 1.5 + 0.3).bfloat16().float()
    mra = stats_ref(xa)
    w1 = _rng_t(seed + 3, (Cout, Ca, 3, 3, 3), 1.0 / math.sqrt(27 * Ca))
    ws = _rng_t(seed + 4, (Cout, Ca, 3, 3, 3), 1.0 / math.sqrt(27 * Ca))
    wcat = torch.cat([w1, ws], 0).bfloat16().double()
    old = {k: os.environ.get(k) for k in ('RSUPER_S2K', 'RSUPER_S2D')}
    os.environ['RSUPER_S2K'] = os.environ['RSUPER_S2D'] = kernel
    try:
        sa = ops.Src(to_cl(xa, dt), mr=mra.to(DEV))
        if kind == 'fwd':
            xh = _xhat_bf16(xa, mra)
            ref = F.conv3d(xh, wcat, stride=2, padding=1)
            mag = F.conv3d(xh.abs(), wcat.abs(), stride=2, padding=1)
            nc, cnt, K = 2 * Cout, OD * OH * OW, 27 * Ca
            wp = ops.pack_weights(dt, 0, w1.to(DEV), ws.to(DEV), Ca, 0, Cout, Cout, 64)
            out = torch.full((N, OD, OH, OW, nc), float('nan'), device=DEV, dtype=dt)
            part = torch.full((N, ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 1, Ca, 0, nc, N, D, H, W), nc, 2), float('nan'), device=DEV, dtype=torch.float32)
            ops.igemm_s2(1, sa, None, wp, nc, dims, out, part)
            fin = ops.stats_finalize(part, cnt)
        else:
            dy = _rng_t(seed + 6, (N, 2 * Cout, OD, OH, OW)).bfloat16().float()
            opad = (1 - D % 2, 1 - H % 2, 1 - W % 2)
            ref = F.conv_transpose3d(dy.double(), wcat, stride=2, padding=1, output_padding=opad)
            mag = F.conv_transpose3d(dy.double().abs(), wcat.abs(), stride=2, padding=1, output_padding=opad)
            mask = (xa > mra[:, :, 0].float()[:, :, None, None, None]).double()
            ref, mag = ref * mask, mag * mask
            nc, cnt, K = Ca, D * H * W, 27 * 2 * Cout
            wp = ops.pack_weights(dt, 1, w1.to(DEV), ws.to(DEV), Cout, Cout, Ca, 0, 64)
            out = torch.full((N, D, H, W, Ca), float('nan'), device=DEV, dtype=dt)
            part = torch.full((N, ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 2, Cout, Cout, Ca, N, D, H, W), Ca, 2), float('nan'), device=DEV, dtype=torch.float32)
            dcl = to_cl(dy, dt)
            ops.igemm_s2(2, ops.Src(dcl, C=Cout), ops.Src(dcl, C=Cout, off=Cout), wp, Ca, dims, out, part, ea=sa)
            fin = ops.stats_finalize(part, cnt, mode=1)
        torch.cuda.synchronize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    got = from_cl(out).double()
    bound = 2.0 ** -8 * ref.abs() + (K / 8 + 4) * 2.0 ** -24 * mag + 1e-30
    ratio = (got - ref).abs() / bound
    ratio = torch.where(torch.isfinite(ratio), ratio, torch.full_like(ratio, float('inf')))
    fin = fin.cpu().double()
    rms = (got * got).mean(dim=(2, 3, 4)).sqrt().clamp_min(1e-30)
    if kind == 'fwd':
        mean = got.mean(dim=(2, 3, 4)); var = (got * got).mean(dim=(2, 3, 4)) - mean * mean
        e_s = max(((fin[..., 0] - mean).abs() / rms).max().item(), (fin[..., 1] * torch.sqrt(var.clamp_min(0) + 1e-4) - 1.0).abs().max().item())
    else:
        xn = (xa.double() - mra[:, :, 0].double()[:, :, None, None, None]) * mra[:, :, 1].double()[:, :, None, None, None]
        e_s = max(((fin[..., 0] - got.mean(dim=(2, 3, 4))).abs() / rms).max().item(), ((fin[..., 1] - (got * xn).mean(dim=(2, 3, 4))).abs() / rms).max().item())
    e_s = e_s if math.isfinite(e_s) else float('inf')
    return result(f'conv_exact_s2[{kind} kernel{kernel} N{N} S{S} {Ca}->2x{Cout}]', max(ratio.max().item(), e_s / 1e-4), 1.0,
                  f'max |got-ref| / (1 bf16 rounding + f32 slack) {ratio.max().item():.3f}; emitted rows vs float64 sums of the written tensor {e_s:.1e} (bound 1e-4)')


def check_wgrad_xhat(N, S, Ca, Cb, Ya, Yb, seed=0):
    """Weight gradient on PRE-NORMALISED bf16 sources (no statistics: csrc/conv3d_wgrad_dma.hip, operands by LDS-DMA) against the float64
    gradient of F.conv3d on exactly the bf16 operands the kernel sees -- f32 accumulation is the only difference, hence the tight bound."""
    from rsuper_amd.hip import ops
    D, H, W = S
    dt = torch.bfloat16
    xa = torch.relu(_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3).bfloat16()
    xb = torch.relu(_rng_t(seed + 2, (N, Cb, D, H, W)) * 1.5 - 0.2).bfloat16() if Cb else None
    dy1 = _rng_t(seed + 6, (N, Ya, D, H, W)).bfloat16()
    dy2 = _rng_t(seed + 7, (N, Yb, D, H, W)).bfloat16() if Yb else None
    xh = (xa if xb is None else torch.cat([xa, xb], 1)).double()
    refs = []
    for dy, Y in ((dy1, Ya), (dy2, Yb)):
        if dy is None:
            continue
        w = torch.zeros((Y, Ca + Cb, 3, 3, 3), dtype=torch.float64, requires_grad=True)
        F.conv3d(xh, w, padding=1).backward(dy.double())
        refs.append(w.grad.float())
    sa = ops.Src(to_cl(xa.float(), dt))
    sb = ops.Src(to_cl(xb.float(), dt)) if Cb else None
    dw1 = torch.full((Ya, Ca + Cb, 3, 3, 3), float('nan'), device=DEV)
    dw2 = torch.full((Yb, Ca + Cb, 3, 3, 3), float('nan'), device=DEV) if Yb else None
    ops.wgrad(sa, sb, ops.Src(to_cl(dy1.float(), dt)), ops.Src(to_cl(dy2.float(), dt)) if Yb else None, dw1, dw2, (N, D, H, W))
    torch.cuda.synchronize()
    e = relerr(dw1.cpu(), refs[0])
    if Yb:
        e = max(e, relerr(dw2.cpu(), refs[1]))
    return result(f'wgrad_xhat[bf16 N{N} S{S} {Ca}+{Cb}->{Ya}+{Yb}]', e, 1e-4, f'dw {e:.2e}')


def check_wgrad_s2(mode, N, S, Ca, Ya, Yb, seed=0):
    """Strided weight gradient (csrc/conv3d_wgrad_s2.hip) of [conv1 | shortcut] against autograd of F.conv3d(stride=2, padding=1) on the
    normalised + rectified input (float64 reference of the operands the kernel sees)."""
    from rsuper_amd.hip import ops
    import torch.nn.functional as F
    dt = DT[mode]
    D, H, W = S
    OD, OH, OW = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    xa = rnd(_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3, mode)
    dy1 = rnd(_rng_t(seed + 6, (N, Ya, OD, OH, OW)), mode)
    dy2 = rnd(_rng_t(seed + 7, (N, Yb, OD, OH, OW)), mode) if Yb else None
    mr = stats_ref(xa)
    xh = torch.relu((xa.double() - mr[:, :, 0].double()[:, :, None, None, None]) * mr[:, :, 1].double()[:, :, None, None, None])
    if mode == 'bf16':
        xh = xh.float().bfloat16().double()
    refs = []
    for dy, Y in ((dy1, Ya), (dy2, Yb)):
        if dy is None:
            continue
        w = torch.zeros((Y, Ca, 3, 3, 3), dtype=torch.float64, requires_grad=True)
        F.conv3d(xh, w, stride=2, padding=1).backward(dy.double())
        refs.append(w.grad.float())
    sa = ops.Src(to_cl(xa, dt), mr=mr.to(DEV))
    dw1 = torch.full((Ya, Ca, 3, 3, 3), float('nan'), device=DEV)
    dw2 = torch.full((Yb, Ca, 3, 3, 3), float('nan'), device=DEV) if Yb else None
    ops.wgrad_s2(sa, ops.Src(to_cl(dy1, dt)), ops.Src(to_cl(dy2, dt)) if Yb else None, dw1, dw2, (N, D, H, W))
    torch.cuda.synchronize()
    e = relerr(dw1.cpu(), refs[0])
    if Yb:
        e = max(e, relerr(dw2.cpu(), refs[1]))
    return result(f'wgrad_s2[{mode} N{N} S{S} {Ca}->{Ya}+{Yb}]', e, TOL[mode], f'dw {e:.2e}')


def check_token_attn(B, L, heads, dh, seed=0):
    """csrc/token_attn.hip against the ATen chain of Attention.forward (trans_layers.py:52-84) in float64, forward and backward."""
    from rsuper_amd.hip import ops
    import torch.nn.functional as F
    qkv = _rng_t(seed + 1, (B, L, 3 * heads * dh))
    go = _rng_t(seed + 2, (B, L, heads * dh))
    scale = dh ** -0.5
    r = qkv.double().requires_grad_(True)
    q, k, v = (t.reshape(B, L, heads, -1).transpose(1, 2) for t in r.chunk(3, -1))
    o_ref = torch.matmul(F.softmax(torch.matmul(q, k.transpose(-1, -2)) * scale, -1), v).transpose(1, 2).reshape(B, L, -1)
    o_ref.backward(go.double())
    x = qkv.to(DEV).requires_grad_(True)
    o = ops.TokenAttnFn.apply(x, heads, scale)
    o.backward(go.to(DEV))
    torch.cuda.synchronize()
    e_o, e_g = relerr(o.detach().cpu(), o_ref.detach()), relerr(x.grad.cpu(), r.grad)
    return result(f'token_attn[B{B} L{L} h{heads} d{dh}]', max(e_o, e_g), 2e-5, f'o {e_o:.2e} dqkv {e_g:.2e}')


def check_tr16_probe():
    """Raw ds_read_b64_tr_b16 behaviour is exercised through wgrad with tr=1 vs tr=0 (bit-identical operands ->
    identical MFMA results up to atomic order)."""
    r0 = check_conv_bwd('bf16', 1, (4, 4, 16), 32, 0, 32, False, seed=11, tr=0)
    r1 = check_conv_bwd('bf16', 1, (4, 4, 16), 32, 0, 32, False, seed=11, tr=1)
    return result('wgrad_tr16_vs_u16', abs(r1['err'] - r0['err']) + (0 if r1['ok'] else 1), 1e-3, f"tr0 {r0['note']} | tr1 {r1['note']}")


# ================================================================================================ pool / upsample / stem / head
def check_pool(mode):
    from rsuper_amd.hip import ops
    dt = DT[mode]
    g = golden('blocks')
    x = rnd(T(synth.rng(61).standard_normal((1, 8, 8, 8, 8)).astype(np.float32)), mode).requires_grad_(True)
    y_ref = F.max_pool3d(x, 2)
    go = rnd(T(synth.rng(62).standard_normal(tuple(y_ref.shape)).astype(np.float32)), mode)
    y_ref.backward(go)
    xc = to_cl(x.detach(), dt).requires_grad_(True)
    y, mr = ops.MaxPoolFn.apply(xc)
    y.backward(to_cl(go, dt))
    torch.cuda.synchronize()
    e = max(relerr(from_cl(y.detach()), y_ref.detach()), relerr(from_cl(xc.grad), x.grad),
            relerr(mr.cpu()[..., 0], stats_ref(y_ref.detach())[..., 0]))
    if mode == 'f32':
        e = max(e, relerr(from_cl(y.detach()), T(g['pool_y'])), relerr(from_cl(xc.grad), T(g['pool_dx'])))
    return result(f'maxpool[{mode}]', e, 1e-6 if mode == 'f32' else 1e-2)


def check_pool_skip(mode, shape=(2, 8, 6, 10, 16), seed=64):
    """MaxPoolSkipFn (pooling layer whose input is also a skip connection: both gradients summed inside the max-pool backward kernel) against the
    fp32 reference autograd of `F.max_pool3d(x, 2)` + a second use of x, and bit for bit against MaxPoolFn + autograd's own accumulation; even sizes
    (fused kernel) and odd ones (two steps).  shape = (N, D, H, W, C)."""
    from rsuper_amd.hip import ops
    dt = DT[mode]
    N, D, H, W, C = shape
    x = rnd(T(synth.rng(seed).standard_normal((N, C, D, H, W)).astype(np.float32)), mode).requires_grad_(True)
    y_ref = F.max_pool3d(x, 2)
    go = rnd(T(synth.rng(seed + 1).standard_normal(tuple(y_ref.shape)).astype(np.float32)), mode)
    gs = rnd(T(synth.rng(seed + 2).standard_normal(tuple(x.shape)).astype(np.float32)), mode)
    (y_ref * go).sum().backward(retain_graph=True)
    x.grad += gs                                                     # the skip path's gradient
    xa = to_cl(x.detach(), dt).requires_grad_(True)
    y, mr, skip = ops.MaxPoolSkipFn.apply(xa)
    torch.autograd.backward([y, skip], [to_cl(go, dt), to_cl(gs, dt)])
    xb = to_cl(x.detach(), dt).requires_grad_(True)
    y2, _ = ops.MaxPoolFn.apply(xb)
    torch.autograd.backward([y2, xb * 1.0], [to_cl(go, dt), to_cl(gs, dt)])
    torch.cuda.synchronize()
    same = bool(torch.equal(xa.grad, xb.grad)) and bool(torch.equal(y, y2))
    e = max(relerr(from_cl(y.detach()), y_ref.detach()), relerr(from_cl(xa.grad), rnd(x.grad, mode)))
    return result(f'maxpool_skip[{mode} {shape}]', e if same else 1.0, 1e-6 if mode == 'f32' else 1e-2, f'identical to MaxPoolFn + accumulation: {same}')


def check_upsample(mode, Cin=8, I=3, O=6, seed=63):
    from rsuper_amd.hip import ops
    dt = DT[mode]
    x = rnd(T(synth.rng(seed).standard_normal((1, Cin, I, I, I)).astype(np.float32)), mode).requires_grad_(True)
    y_ref = F.interpolate(x, size=(O, O, O), mode='trilinear', align_corners=True)
    go = rnd(T(synth.rng(seed + 1).standard_normal(tuple(y_ref.shape)).astype(np.float32)), mode)
    y_ref.backward(go)
    xc = to_cl(x.detach(), dt).requires_grad_(True)
    y, mr = ops.UpsampleFn.apply(xc, (O, O, O))
    y.backward(to_cl(go, dt))
    torch.cuda.synchronize()
    e = max(relerr(from_cl(y.detach()), y_ref.detach()), relerr(from_cl(xc.grad), x.grad))
    e2 = relerr(mr.cpu()[..., 1], stats_ref(rnd(y_ref.detach(), mode))[..., 1])
    return result(f'upsample[{mode} {I}->{O} C{Cin}]', max(e, e2 * 0.1), 2e-5 if mode == 'f32' else 2e-2, f'val/grad {e:.2e} rstd {e2:.2e}')


def check_upsample_bwd(N, I, O, C, seed=80):
    """rsuper_upsample_bwd (bf16: the row-sharing kernel, csrc/unet_misc.hip upsample_bwd4_kernel) on a random gradient against a float64 evaluation
    (autograd of F.interpolate) at the UNet's shapes scaled down, odd / ragged sizes, scale 1 and ~3 (the latter falls to the gather kernel).
    I / O: (d, h, w) sizes.  Tolerance: the bf16 rounding of the result."""
    from rsuper_amd.hip import ops
    from rsuper_amd.hip import lib as _lib
    dt = torch.bfloat16
    dy = T(synth.rng(seed).standard_normal((N,) + tuple(O) + (C,)).astype(np.float32)).to(DEV).to(dt)
    got = torch.full((N,) + tuple(I) + (C,), float('nan'), device=DEV, dtype=dt)
    _lib.check(ops._L().rsuper_upsample_bwd(1, ops._ptr(dy), C, ops._ptr(got), C, N, *I, *O, C, ops._stream()), 'upsample_bwd')
    torch.cuda.synchronize()
    xin = torch.zeros((N, C) + tuple(I), dtype=torch.float64, requires_grad=True)
    F.interpolate(xin, size=tuple(O), mode='trilinear', align_corners=True).backward(dy.cpu().double().permute(0, 4, 1, 2, 3).contiguous())
    e = relerr(got.float().cpu(), xin.grad.permute(0, 2, 3, 4, 1).float())
    return result(f'upsample_bwd[bf16 N{N} {I}->{O} C{C}]', e, 1e-2)


def check_unet_wide(mode, base=64, classes=70, S=32):
    """Widths beyond the shipped configuration (base_ch 64: 640 channels at the bottom, 64-channel stem / head in two 32-channel passes of the small
    weight-gradient kernel; 70 classes: three 32-class passes) -- the reference UNet takes any width (rsuper_train/model/dim3/unet.py:31-47).
    Logits against the CPU oracle, every gradient finite; stem / head gradients are pinned by check_stem / check_head at these widths."""
    from rsuper_amd.model.dim3.unet import UNet
    net = UNet(1, base, num_classes=classes, compute_dtype=mode)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 5)
    net.load_state_dict({k: T(v) for k, v in sd.items()})
    net = net.to(DEV)
    img = T(synth.image(1, S, seed=9))
    y = net(img.to(DEV))['segmentation']
    y.square().mean().backward()
    torch.cuda.synchronize()
    ref = uo.unet_forward({k: T(v) for k, v in sd.items()}, img)
    e = relerr(y.detach().cpu(), ref)
    fin = all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    tol = 1e-3 if mode == 'f32' else 0.3
    return result(f'unet_wide[{mode} base {base}, {classes} classes, {S}^3]', e if fin else 1e9, tol, f'logits vs oracle {e:.2e}; gradients finite: {fin}')


def check_stem(mode, C=8, S=12):
    from rsuper_amd.hip import ops
    dt = DT[mode]
    img = T(synth.image(2, S, seed=5))
    w = _rng_t(21, (C, 1, 3, 3, 3), 0.2).requires_grad_(True)
    y_ref = F.conv3d(img, w, padding=1)
    go = rnd(_rng_t(22, tuple(y_ref.shape)), mode)
    y_ref.backward(go)
    wd = w.detach().to(DEV).requires_grad_(True)
    y, mr = ops.StemFn.apply(img.to(DEV), wd, dt)
    y.backward(to_cl(go, dt))
    torch.cuda.synchronize()
    e = max(relerr(from_cl(y.detach()), y_ref.detach()), relerr(wd.grad.cpu(), w.grad),
            relerr(mr.cpu()[..., 1], stats_ref(rnd(y_ref.detach(), mode))[..., 1]) * 0.1)
    return result(f'stem[{mode} C{C}]', e, 1e-5 if mode == 'f32' else 1e-2)


def check_head(mode, C=8, K=5, S=10):
    from rsuper_amd.hip import ops
    dt = DT[mode]
    x = rnd(_rng_t(31, (2, C, S, S, S)), mode).requires_grad_(True)
    w = _rng_t(32, (K, C, 1, 1, 1), 0.3).requires_grad_(True)
    b = _rng_t(33, (K,), 0.1).requires_grad_(True)
    y_ref = F.conv3d(x, w, b)
    go = _rng_t(34, tuple(y_ref.shape))
    y_ref.backward(go)
    xc = to_cl(x.detach(), dt).requires_grad_(True)
    wd, bd = w.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    y = ops.HeadFn.apply(xc, wd, bd)
    y.backward(go.to(DEV))
    torch.cuda.synchronize()
    e = max(relerr(y.detach().cpu(), y_ref.detach()), relerr(from_cl(xc.grad), x.grad), relerr(wd.grad.cpu(), w.grad),
            relerr(bd.grad.cpu(), b.grad))
    return result(f'head[{mode} C{C} K{K}]', e, 1e-5 if mode == 'f32' else 1e-2)


# ================================================================================================ blocks / UNet vs golden
def _load_block(tag, ci, co, seed, mode):
    from rsuper_amd.model.dim3.conv_layers import BasicBlock
    blk = BasicBlock(ci, co, stride=2) if tag.endswith('_s2') else BasicBlock(ci, co)
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict({k: T(v) for k, v in synth.fill_state_dict(shapes, seed).items()})
    return blk.to(DEV)


def check_basic_block(mode, tag, ci, co, S, seed):
    g = golden('blocks')
    dt = DT[mode]
    blk = _load_block(tag, ci, co, seed, mode)
    x = T(synth.rng(40 + ci).standard_normal((2, ci, S, S, S)).astype(np.float32))
    So = (S + 1) // 2 if tag.endswith('_s2') else S
    go = T(synth.rng(50 + co).standard_normal((2, co, So, So, So)).astype(np.float32))
    from rsuper_amd.hip import ops
    xc = to_cl(x, dt).requires_grad_(True)
    mr = stats_ref(rnd(x, mode)).to(DEV)
    y, _ = blk(xc, mr)
    y.backward(to_cl(go, dt))
    torch.cuda.synchronize()
    e_y = err_for(mode, from_cl(y.detach()), T(g[f'{tag}_y']))
    e_dx = err_for(mode, from_cl(xc.grad), T(g[f'{tag}_dx']))
    e_w = max(err_for(mode, p.grad.cpu(), T(g[f'{tag}_dw_{k}'])) for k, p in blk.named_parameters())
    tol = 2e-4 if mode == 'f32' else 8e-2      # bf16: relative L2; ~0.4 % of ReLU masks flip -> ~5 % in gradients
    return result(f'basic_block_golden[{mode} {tag}]', max(e_y, e_dx, e_w), tol, f'y {e_y:.2e} dx {e_dx:.2e} dw {e_w:.2e}')


def make_tiny_unet(mode, seed=3, classes=None):
    from rsuper_amd.model.dim3.unet import UNet
    classes = classes or synth.TINY_CLASSES
    net = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=mode)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: T(v) for k, v in synth.fill_state_dict(shapes, seed).items()})
    return net.to(DEV)


def f64_tiny_grads(pool):
    """Gradients of the tiny UNet from the float64 CPU restatement (strided 4k subsample per tensor).  The fp32 reference fixture itself
    is up to 1.4e-2 of max away from these (profiles/r02_tiny_grad_noise.txt), so fp32 kernels are held to the float64 values."""
    shapes = uo.unet_param_shapes(1, 8, len(synth.TINY_CLASSES), pool=pool)
    sd = {k: T(v).double().requires_grad_(True) for k, v in synth.fill_state_dict(shapes, 3).items()}
    y = uo.unet_forward(sd, T(synth.image(1, 48, seed=1234)).double(), pool=pool)
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    y.backward(T(go).double())
    return {k: synth.subsample(v.grad.numpy(), 4096)[0] for k, v in sd.items()}


def check_unet_tiny_nopool(mode):
    """UNet(..., pool=False): strided down-sampling (unet_utils.py:38-39) against the reference fixture."""
    from rsuper_amd.model.dim3.unet import UNet
    g = golden('unet_tiny')
    net = UNet(1, 8, num_classes=len(synth.TINY_CLASSES), block='BasicBlock', norm='in', pool=False, compute_dtype=mode)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: T(v) for k, v in synth.fill_state_dict(shapes, 3).items()})
    net = net.to(DEV)
    y = net(T(synth.image(1, 48, seed=1234)).to(DEV))['segmentation']
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    y.backward(T(go).to(DEV))
    torch.cuda.synchronize()
    sub, _ = synth.subsample(y.detach().cpu().numpy(), 8192)
    e_y = err_for(mode, T(sub), T(g['nopool_logits_sub']))
    worst, wk, worst64 = 0.0, '', 0.0
    g64 = f64_tiny_grads(False) if mode == 'f32' else None
    for k, p in net.named_parameters():
        gsub, _ = synth.subsample(p.grad.cpu().numpy(), 4096)        # element-wise over a strided sample of the whole tensor
        sc = max(g[f'nopool_g_{k}_summary'][2], 1e-12)
        e = float(np.abs(gsub - g[f'nopool_g_{k}_sub']).max() / sc) if mode == 'f32' else l2err(T(gsub), T(g[f'nopool_g_{k}_sub']))
        if e > worst:
            worst, wk = e, k
        if g64 is not None:
            worst64 = max(worst64, float(np.abs(gsub - g64[k]).max() / sc))
    # f32: <= 1.2e-2 of max against the float64 restatement (measured 9.3e-3; the fp32 reference is 1.1e-2 away from it) and <= 2e-2 against
    # the fp32 reference fixture (measured 1.1e-2, i.e. the reference's own rounding noise)
    # bf16: relative L2 per tensor; the stem gradient of this ill-conditioned net is mostly rounding noise in bf16 (measured 0.55 at
    # inc.conv1, 0.04 on the logits), so bf16 only guards against gross errors here -- per-layer bf16 parity is checked block by block
    tol_y, tol_g = (1e-4, 2e-2) if mode == 'f32' else (0.25, 0.8)          # f32 logits: measured 6.5e-6
    return result(f'unet_tiny_nopool[{mode}]', max(e_y / tol_y, worst / tol_g, worst64 / 1.2e-2), 1.0,
                  f'logits {e_y:.2e} (tol {tol_y}); worst grad vs reference {worst:.2e} @ {wk} (tol {tol_g}); vs float64 {worst64:.2e} (tol 1.2e-2)')


def check_unet_tiny(mode):
    g = golden('unet_tiny')
    net = make_tiny_unet(mode)
    img = T(synth.image(1, 48, seed=1234)).to(DEV)
    y = net(img)['segmentation']
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    y.backward(T(go).to(DEV))
    torch.cuda.synchronize()
    sub, step = synth.subsample(y.detach().cpu().numpy(), 8192)
    ref = g['logits_sub']
    e_y = err_for(mode, T(sub), T(ref))
    e_abs = float(np.abs(sub - ref).max())
    worst, wk, worst64 = 0.0, '', 0.0
    g64 = f64_tiny_grads(True) if mode == 'f32' else None
    for k, p in net.named_parameters():
        r = g[f'g_{k}_summary']
        sc = max(r[2], 1e-12)
        gsub, _ = synth.subsample(p.grad.cpu().numpy(), 4096)        # strided over the whole tensor, not the first 64 entries
        if mode == 'f32':
            e = float(np.abs(gsub - g[f'g_{k}_sub']).max() / sc)
            worst64 = max(worst64, float(np.abs(gsub - g64[k]).max() / sc))
        else:
            e = l2err(T(gsub), T(g[f'g_{k}_sub']))
        if e > worst:
            worst, wk = e, k
    if mode == 'f32':
        # gradients: <= 1e-2 of max against the float64 restatement (measured 7.5e-3) -- the fp32 reference fixture is itself 1.4e-2 away
        # from float64 on this deep pre-activation chain (profiles/r02_tiny_grad_noise.txt), hence 2e-2 against the fixture
        tol_y, tol_g, tol_64 = 1e-4, 2e-2, 1e-2
        ok_err = max(e_y / tol_y, worst / tol_g, worst64 / tol_64)
        return result(f'unet_tiny_golden[{mode}]', ok_err, 1.0,
                      f'logits rel {e_y:.2e} abs {e_abs:.2e} (tol {tol_y}); worst grad vs reference {worst:.2e} @ {wk} (tol {tol_g}); '
                      f'vs float64 restatement {worst64:.2e} (tol {tol_64})')
    # bf16: the oracle is the bf16-rounding-emulating CPU restatement (oracle/unet_oracle.py, emulate_bf16=True); the distance
    # to the fp32 golden (~0.11) is inherent to bf16 on this net (tests/test_oracle_vs_golden.py::test_bf16_emulation_...).
    sd = {k: T(v) for k, v in synth.fill_state_dict(uo.unet_param_shapes(1, 8, len(synth.TINY_CLASSES)), 3).items()}
    with torch.no_grad():
        y_emu = uo.unet_forward(sd, T(synth.image(1, 48, seed=1234)), emulate_bf16=True)
    e_emu = l2err(y.detach().cpu(), y_emu)
    gfin = all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    # two valid bf16 executions (different fp32 accumulation orders -> different rounding decisions) differ by ~6e-2 here,
    # i.e. by about as much as either differs from fp32: the bound is the net's conditioning, kernels are checked per layer.
    return result(f'unet_tiny_bf16_vs_emulated_oracle', e_emu if gfin else float('inf'), 0.08,
                  f'logits L2 vs bf16-emulating oracle {e_emu:.2e} (tol 0.08); vs fp32 golden {e_y:.2e}; worst grad L2 vs fp32 golden '
                  f'{worst:.2e} @ {wk}; grads finite {gfin}')


# ================================================================================================ losses
def check_cnorm(C, dims, relu, N=2, seed=95):
    """Stand-alone InstanceNorm [+ ReLU] (csrc/instnorm.hip) forward / backward against F.instance_norm (+ F.relu) on the CPU."""
    from rsuper_amd.hip import ops
    D, H, W = dims
    x = (_rng_t(seed, (N, C, D, H, W)) * 1.7 + 0.3).requires_grad_(True)
    y_ref = F.instance_norm(x, eps=1e-5)
    y_ref = F.relu(y_ref) if relu else y_ref
    go = _rng_t(seed + 1, tuple(y_ref.shape))
    y_ref.backward(go)
    xd = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    y = ops.ChannelNormFn.apply(xd, 1e-5, relu)
    y.backward(go.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    torch.cuda.synchronize()
    e = max(relerr(y.detach().cpu().permute(0, 4, 1, 2, 3), y_ref.detach()), relerr(xd.grad.cpu().permute(0, 4, 1, 2, 3), x.grad))
    return result(f'cnorm C{C} {dims} relu{int(relu)}', e, 2e-5)


def check_depthwise(C, dims, N=2, seed=91):
    """Depthwise 3x3x3 convolution (csrc/depthwise.hip) forward, data gradient and weight gradient against F.conv3d(groups=C) on the
    CPU (fp32; 1e-5 of max: pure fp32 multiply-adds in a different summation order)."""
    from rsuper_amd.hip import ops
    D, H, W = dims
    x = _rng_t(seed, (N, C, D, H, W)).requires_grad_(True)
    w = _rng_t(seed + 1, (C, 1, 3, 3, 3), 0.3).requires_grad_(True)
    y_ref = F.conv3d(x, w, padding=1, groups=C)
    go = _rng_t(seed + 2, tuple(y_ref.shape))
    y_ref.backward(go)
    xd = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    y = ops.DepthwiseConvFn.apply(xd, wd)
    y.backward(go.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    torch.cuda.synchronize()
    e = max(relerr(y.detach().cpu().permute(0, 4, 1, 2, 3), y_ref.detach()), relerr(xd.grad.cpu().permute(0, 4, 1, 2, 3), x.grad),
            relerr(wd.grad.cpu(), w.grad))
    return result(f'depthwise3 C{C} {dims}', e, 1e-5)


def check_squeeze_excite(C, dims, N=2, seed=93):
    """SEBlock (csrc/instnorm.hip statistics + affine apply around the ATen excitation) forward and all five gradients against the
    float64 composition x * sigmoid(W2 relu(W1 mean(x) + b1) + b2) on the CPU."""
    from rsuper_amd.hip import ops
    r = C // 4
    x = _rng_t(seed, (N, *dims, C)).requires_grad_(True)
    ws = [_rng_t(seed + 1, (r, C, 1, 1, 1), 0.3), _rng_t(seed + 2, (r,), 0.3), _rng_t(seed + 3, (C, r, 1, 1, 1), 0.3), _rng_t(seed + 4, (C,), 0.3)]
    ws = [w.requires_grad_(True) for w in ws]
    go = _rng_t(seed + 5, (N, *dims, C))
    xd, wd = x.double(), [w.double() for w in ws]
    sc = torch.sigmoid(F.linear(F.relu(F.linear(xd.mean((1, 2, 3)), wd[0].flatten(1), wd[1])), wd[2].flatten(1), wd[3]))
    y_ref = xd * sc[:, None, None, None, :]
    (y_ref * go.double()).sum().backward()
    xg = x.detach().to(DEV).requires_grad_(True)
    wg = [w.detach().to(DEV).requires_grad_(True) for w in ws]
    y = ops.SqueezeExciteFn.apply(xg, *wg)
    (y * go.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    e = max([relerr(y.detach().cpu(), y_ref.detach().float()), relerr(xg.grad.cpu(), x.grad)] + [relerr(a.grad.cpu(), b.grad) for a, b in zip(wg, ws)])
    return result(f'squeeze_excite C{C} {dims}', e, 2e-5)


def check_linear_splitk(rows, cin, cout, bias, seed=99):
    """1x1x1 convolution of the attention stages with the voxel-split weight gradient (model/dim3/medformer_utils.py linear) against
    F.linear in float64 on the CPU: output and the three gradients."""
    from rsuper_amd.model.dim3 import medformer_utils as mu
    x = _rng_t(seed, (2, rows // 2, cin)).requires_grad_(True)
    w = _rng_t(seed + 1, (cout, cin), 0.2).requires_grad_(True)
    b = _rng_t(seed + 2, (cout,)).requires_grad_(True) if bias else None
    go = _rng_t(seed + 3, (2, rows // 2, cout))
    y_ref = F.linear(x.double(), w.double(), None if b is None else b.double())
    (y_ref * go.double()).sum().backward()
    xd, wd = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    bd = b.detach().to(DEV).requires_grad_(True) if bias else None
    assert rows >= mu.SPLITK_MIN_ROWS
    prev, was = torch.backends.cuda.preferred_blas_library(), mu.gemm_library.active
    mu.gemm_library.active = True                       # the per-call library switch of the opt-in rocBLAS mode (RSUPER_MF_BLAS=cublas)
    torch.backends.cuda.preferred_blas_library('cublas')
    try:
        y = mu.linear(xd, wd, bd)
        (y * go.to(DEV)).sum().backward()
        torch.cuda.synchronize()
    finally:                                            # process-wide setting: leave it as it was for the checks that follow
        mu.gemm_library.active = was
        torch.backends.cuda.preferred_blas_library(prev)
    errs = [relerr(y.detach().cpu(), y_ref.detach().float()), relerr(xd.grad.cpu(), x.grad), relerr(wd.grad.cpu(), w.grad)]
    if bias:
        errs.append(relerr(bd.grad.cpu(), b.grad))
    return result(f'linear_splitk rows{rows} {cin}->{cout} bias{int(bias)}', max(errs), 2e-5)


def check_pointwise(mode, R, K, N, bias, seed=103):
    """csrc/pointwise.hip (1x1x1 convolution / linear layer as an MFMA GEMM, forward and data gradient) against float64 matmul:
    exact-f32 MFMA at 2e-6 of the output scale, bf16 operands at 1.5e-2 (fp32 accumulate; measured 2e-3)."""
    from rsuper_amd.hip import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(R, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) if bias else None
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    dy = torch.randn(R, N, generator=g)
    refd = dy.double() @ w.double()
    comp = DT[mode]
    refw, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    y = ops.pointwise_gemm(x.to(DEV), w.to(DEV), b.to(DEV) if bias else None, 0, comp)
    dx = ops.pointwise_gemm(dy.to(DEV), w.to(DEV), None, 1, comp)
    dw, db = ops.pointwise_wgrad(dy.to(DEV), x.to(DEV), bias, comp)
    dw2, db2 = ops.pointwise_wgrad(dy.to(DEV), x.to(DEV), bias, comp)
    torch.cuda.synchronize()
    e = max(relerr(y.cpu(), ref), relerr(dx.cpu(), refd), relerr(dw.cpu(), refw))
    if bias:                                   # column sums: fp32 adds of the UNROUNDED dy in both modes
        e = max(e, relerr(db.cpu(), refb) * (1.0 if mode == 'f32' else 1e3))
    det = bool(torch.equal(dw, dw2)) and (not bias or bool(torch.equal(db, db2)))
    # shortcut in the epilogue: the same fp32 additions in the same order as a separate add
    res = torch.randn(R, N, generator=g).to(DEV)
    yr = ops.pointwise_gemm(x.to(DEV), w.to(DEV), b.to(DEV) if bias else None, 0, comp, res)
    det = det and bool(torch.equal(yr, y + res))
    # batched fragment packing (ops.pointwise_prepack): a parameter the GEMM has seen once is packed by the batch launch from then on -- same bits;
    # and a raw-pointer update of the parameter (WEIGHTS_EPOCH) must not be served stale fragments
    wp = torch.nn.Parameter(w.to(DEV))
    y1 = ops.pointwise_gemm(x.to(DEV), wp, None, 0, comp)
    ops.pointwise_prepack(comp)
    hit = (wp.data_ptr(), 0, ops._DT[comp]) in ops._PW_PACKED
    y2 = ops.pointwise_gemm(x.to(DEV), wp, None, 0, comp)
    with torch.no_grad():
        wp.mul_(2.0)
    y3 = ops.pointwise_gemm(x.to(DEV), wp, None, 0, comp)
    ops.WEIGHTS_EPOCH += 1
    y4 = ops.pointwise_gemm(x.to(DEV), wp, None, 0, comp)
    det = det and hit and bool(torch.equal(y1, y2)) and bool(torch.equal(y3, 2.0 * y1)) and bool(torch.equal(y4, y3))
    return result(f'pointwise[{mode} R{R} K{K} N{N} bias{int(bias)}]', e if det else float('inf'), 2e-6 if mode == 'f32' else 1.5e-2,
                  'forward + data gradient + weight / bias gradient (run twice: bit-identical)')


def check_cl_planar(N, dims, C, K, seed=101):
    """Channels-last -> planar re-layout (csrc/instnorm.hip cl_planar_kernel) and its gradient: pure data movement, bit-exact against
    x[..., :K].permute(0, 4, 1, 2, 3) and the zero-padded inverse."""
    from rsuper_amd.hip import ops
    x = _rng_t(seed, (N, *dims, C)).to(DEV).requires_grad_(True)
    go = _rng_t(seed + 1, (N, K, *dims)).to(DEV)
    y = ops.PlanarFn.apply(x, K)
    y.backward(go)
    torch.cuda.synchronize()
    y_ref = x.detach()[..., :K].permute(0, 4, 1, 2, 3)
    g_ref = F.pad(go.permute(0, 2, 3, 4, 1), (0, C - K))
    exact = bool(torch.equal(y.detach(), y_ref)) and bool(torch.equal(x.grad, g_ref))
    return result(f'cl_planar N{N} {dims} C{C} K{K}', 0.0 if exact else 1.0, 0.5)


def check_battn(B, L, T, heads, dh, seed=97):
    """Bidirectional attention core (csrc/battn.hip) forward + backward against the einsum / soft-max composition of the reference
    (medformer_utils.py:66-86) evaluated in float64 on the CPU.  fp32 kernels with a different summation order: 2e-5 of max."""
    from rsuper_amd.hip import ops
    inner = heads * dh
    fqv = _rng_t(seed, (B, L, 2 * inner)).requires_grad_(True)
    mqv = _rng_t(seed + 1, (B, T, 2 * inner)).requires_grad_(True)
    gf, gm = _rng_t(seed + 2, (B, L, inner)), _rng_t(seed + 3, (B, T, inner))
    scale = dh ** -0.5

    def split(t):                                           # 'b (dim_head heads) l -> b heads l dim_head' on channels-last rows
        return t.reshape(B, -1, dh, heads).permute(0, 3, 1, 2)

    fq, fv = (split(t) for t in fqv.double().chunk(2, -1))
    mq, mv = (split(t) for t in mqv.double().chunk(2, -1))
    attn = torch.einsum('bhid,bhjd->bhij', fq, mq) * scale                       # voxels x tokens
    f_ref = torch.einsum('bhij,bhjd->bhid', F.softmax(attn, -1), mv).permute(0, 2, 3, 1).reshape(B, L, inner)
    m_ref = torch.einsum('bhji,bhjd->bhid', F.softmax(attn, -2), fv).permute(0, 2, 3, 1).reshape(B, T, inner)
    ((f_ref * gf.double()).sum() + (m_ref * gm.double()).sum()).backward()
    a, b = fqv.detach().to(DEV).requires_grad_(True), mqv.detach().to(DEV).requires_grad_(True)
    f, m = ops.BidirAttnFn.apply(a, b, heads, scale)
    ((f * gf.to(DEV)).sum() + (m * gm.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    e = max(relerr(f.detach().cpu(), f_ref.detach().float()), relerr(m.detach().cpu(), m_ref.detach().float()),
            relerr(a.grad.cpu(), fqv.grad), relerr(b.grad.cpu(), mqv.grad))
    return result(f'battn B{B} L{L} T{T} h{heads} dh{dh}', e, 2e-5)


def check_medformer_tiny(mode):
    """MedFormer forward + backward against the fixture of the reference class (tests/golden/medformer.npz): both heads,
    encoder features / semantic maps (summaries) and a strided sample of every parameter gradient."""
    from rsuper_amd.model.dim3.medformer import MedFormer
    g = golden('medformer')
    cfg = synth.MEDFORMER_TINY
    net = MedFormer(1, len(synth.TINY_CLASSES), compute_dtype=mode, **{k: v for k, v in cfg.items() if k not in ('size', 'seed')})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: T(v) for k, v in synth.fill_state_dict(shapes, cfg['seed']).items()})
    net = net.to(DEV)
    y, aux = net(T(synth.image(1, cfg['size'], seed=1234)).to(DEV))['segmentation']
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    ga = synth.rng(78).standard_normal(tuple(aux.shape)).astype(np.float32) / aux.numel()
    ((y * T(go).to(DEV)).sum() + (aux * T(ga).to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    e_y = err_for(mode, T(synth.subsample(y.detach().cpu().numpy(), 8192)[0]), T(g['logits_sub']))
    e_a = err_for(mode, T(synth.subsample(aux.detach().cpu().numpy(), 8192)[0]), T(g['aux_sub']))
    gmax = max(float(g[f'g_{k}_summary'][2]) for k, _ in net.named_parameters())
    worst, wk = 0.0, ''
    num = den = 0.0
    for k, p in net.named_parameters():
        if p.grad is None:
            return result(f'medformer_tiny[{mode}]', float('inf'), 1.0, f'no gradient for {k}')
        gsub = synth.subsample(p.grad.cpu().numpy(), 1024)[0]
        num += float(np.square(gsub.astype(np.float64) - g[f'g_{k}_sub']).sum())
        den += float(np.square(g[f'g_{k}_sub'].astype(np.float64)).sum())
        # a few tensors have structurally zero gradients (a bias in front of an InstanceNorm): floor the scale at 1e-3 of the largest
        sc = max(g[f'g_{k}_summary'][2], 1e-3 * gmax)
        e = float(np.abs(gsub - g[f'g_{k}_sub']).max() / sc) if mode == 'f32' else float(np.linalg.norm(gsub - g[f'g_{k}_sub']) /
                                                                                         max(np.linalg.norm(g[f'g_{k}_sub']), 1e-3 * gmax * 32))
        if e > worst:
            worst, wk = e, k
    # f32: logits 1e-4 like every f32 parity check.  Gradients: this deep tiny net is ill-conditioned in fp32 -- the fp32 reference fixture
    # is itself 3.2e-2 of max away from the float64 restatement, the HIP path 3.4e-2 (tools/medformer_diag.py) -- hence 6e-2.
    # bf16 (conv stages only; the attention stages stay fp32): loose network-level bounds; per-layer bf16 parity of the conv kernels is
    # checked block by block.
    # The bf16 gradient bound is on ALL gradients together (relative L2 over the sampled entries of every tensor): tensor by tensor the
    # worst one -- a small gradient deep in the attention stages -- is 0.85-1.0 relative in bf16 on this tiny net, i.e. rounding noise of
    # the conv stages, and moves with the GEMM library's summation order; it is reported, not bounded.  All gradients together are 0.59-0.62
    # off in bf16 on this fixture (logits 0.11): a sanity bound only -- the arithmetic of the bf16 conv kernels is pinned block by block, and
    # the f32 mode of the same code path is the parity statement (all gradients 1e-2, logits 2e-5).
    tol_y, tol_g = (1e-4, 6e-2) if mode == 'f32' else (0.15, 0.8)
    allg = (num / max(den, 1e-300)) ** 0.5
    eg = worst if mode == 'f32' else allg
    # f32 also against the FLOAT64 evaluation of the restatement (oracle/medformer_oracle.py, bit-exact against the reference class in fp32):
    # separates kernel error from the rounding noise of the fp32 fixture.  Measured: HIP 4.2e-2 of max from float64, the fp32 reference
    # itself 3.2e-2 (tools/medformer_diag.py) -> bound 5e-2, i.e. within 1.5x of the reference's own distance.
    w64 = 0.0
    if mode == 'f32':
        from oracle import medformer_oracle as mo
        sd = {k: T(v).double().requires_grad_(True) for k, v in synth.fill_state_dict(shapes, cfg['seed']).items()}
        y64, a64 = mo.medformer_forward(sd, T(synth.image(1, cfg['size'], seed=1234)).double(), cfg)
        ((y64 * T(go).double()).sum() + (a64 * T(ga).double()).sum()).backward()
        g64max = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
        for k, p in net.named_parameters():
            f64 = synth.subsample(sd[k].grad.numpy(), 1024)[0]
            w64 = max(w64, float(np.abs(synth.subsample(p.grad.cpu().numpy(), 1024)[0] - f64).max() / max(np.abs(f64).max(), 1e-3 * g64max)))
    return result(f'medformer_tiny[{mode}]', max(e_y / tol_y, e_a / tol_y, eg / tol_g, w64 / 5e-2), 1.0,
                  f'logits {e_y:.2e} aux {e_a:.2e} (tol {tol_y}); worst grad {worst:.2e} @ {wk}; all gradients rel-L2 {allg:.2e} (tol {tol_g} on '
                  f'{"the worst tensor" if mode == "f32" else "all gradients"}); worst grad vs float64 restatement {w64:.2e} (tol 5e-2)')


def check_plane_partials():
    from rsuper_amd.training import losses_foundation as lf
    B, C, S = 2, 3, 12
    V = S ** 3
    g = synth.rng(3)
    x = T(g.standard_normal((B, C, S, S, S)).astype(np.float32) * 3)
    t = T((g.random((B, C, S, S, S)) < 0.3).astype(np.uint8))
    k = T((g.random((B, C, S, S, S)) < 0.8).astype(np.uint8))
    w1 = T(g.random((B, C, S, S, S)).astype(np.float32))
    w2 = T((g.random((B, C, S, S, S)) < 0.5).astype(np.uint8))
    xr = x.clone().requires_grad_(True)
    tf, kf = t.float(), k.float()
    bce = F.binary_cross_entropy_with_logits(xr, tf, reduction='none') * kf
    sg = torch.sigmoid(xr)
    ref = torch.stack([bce.flatten(2).sum(2), (sg * kf).flatten(2).sum(2), (sg * tf * kf).flatten(2).sum(2), (tf * kf).flatten(2).sum(2),
                       (bce * w1).flatten(2).sum(2), (bce * (1 - w2.float())).flatten(2).sum(2)], -1).view(B * C, 6)
    gs = T(g.standard_normal((B * C, 6)).astype(np.float32))
    (ref * gs).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    term = lf._Term(0, V, B * C, t=t.to(DEV), k=k.to(DEV), w1=w1.to(DEV), w2=w2.to(DEV))
    sums, _ = lf._PartialsFn.apply(xd, [term])
    (sums * gs.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    e1 = relerr(sums.detach().cpu(), ref.detach())
    e2 = relerr(xd.grad.cpu(), xr.grad)
    # the same term given the complementary ("unknown") mask with the inverted-k flag must give identical results
    xi = x.to(DEV).requires_grad_(True)
    sums_i, _ = lf._PartialsFn.apply(xi, [lf._Term(0, V, B * C, t=t.to(DEV), k=(1 - k).to(DEV), w1=w1.to(DEV), w2=w2.to(DEV), kinv=True)])
    (sums_i * gs.to(DEV)).sum().backward()
    e3 = max(relerr(sums_i.detach().cpu(), sums.detach().cpu()), relerr(xi.grad.cpu(), xd.grad.cpu()))
    return result('plane_partials', max(e1, e2, e3), 2e-5, f'sums {e1:.2e} grad {e2:.2e} inverted-k {e3:.2e}')


def check_seg_from_sums(B, C, weighted, seed):
    """rsuper_seg_from_sums (loss + Jacobian) against the torch formulation of losses_foundation.py:945-956 / :541-607, including
    classes whose alpha is clamped at 0.2 / 0.8 (no gradient through alpha there) and classes with TP = 0."""
    from rsuper_amd.training import losses_foundation as lf
    g = synth.rng(seed)
    V = 1000
    A = g.random((B, C)) * 400 + 1
    Bs = A * g.random((B, C)) * 0.9
    Cn = Bs + g.random((B, C)) * 300
    A[:, 0] = Bs[:, 0] + 1e-3            # FP ~ 0  -> alpha clamps to 0.2
    Cn[:, 1 % C] = Bs[:, 1 % C]          # FN = 0  -> alpha clamps to 0.8
    if C > 2:
        Bs[:, 2] = 0                     # empty prediction overlap
    S = g.random((B, C)) * 500
    sums = np.zeros((B * C, 6), np.float32)
    sums[:, 0], sums[:, 1], sums[:, 2], sums[:, 3] = S.ravel(), A.ravel(), Bs.ravel(), Cn.ravel()
    sums[:, 4:] = g.random((B * C, 2))
    cw = (g.random((B, C)) + 0.5).astype(np.float32) if weighted else None
    scale = 0.7
    sr = T(sums).double().requires_grad_(True)
    s4 = sr.view(B, C, 6)
    cwt = None if cw is None else T(cw).double()
    bce = ((s4[..., 0] * cwt) if cwt is not None else s4[..., 0]).sum() / float(B * C * V)
    ref = scale * (bce + lf._dice_from_sums(s4[..., 1], s4[..., 2], s4[..., 3], cwt))
    ref.backward()
    sd = T(sums).to(DEV).requires_grad_(True)
    got = lf._SegFromSums.apply(sd, None if cw is None else T(cw).to(DEV), B, C, V, scale)
    (got * 1.5).backward()
    torch.cuda.synchronize()
    e1 = abs(got.item() - ref.item()) / max(abs(ref.item()), 1e-12)
    e2 = relerr(sd.grad.cpu() / 1.5, sr.grad)
    return result(f'seg_from_sums B{B} C{C} w{int(weighted)}', max(e1, e2), 2e-6, f'loss {e1:.2e} jac {e2:.2e}')


def check_dilate():
    from rsuper_amd.hip import ops
    g = synth.rng(11)
    worst = 0
    for shape in [(2, 3, 20, 20, 20), (1, 2, 9, 10, 11), (1, 2, 12, 12, 32), (2, 1, 8, 8, 48)]:
        vol = (g.random(shape) < 0.004).astype(np.uint8)
        vol[0, -1] = 0                  # an empty plane next to occupied ones: the flagged-empty fast path must not leak neighbours in
        for ks in [1, 2, 3, 5, 7, 9, 13, 31]:
            got = ops.dilate_volume(T(vol).to(DEV), ks).cpu().numpy()
            ref = omorph.dilate_volume(vol, ks)
            worst = max(worst, int((got != ref).sum()))
    p = golden('primitives')
    vol = np.unpackbits(p['dil_in'])[:2 * 3 * 8000].reshape(2, 3, 20, 20, 20)
    for ks in [5, 7, 31]:
        got = ops.dilate_volume(T(vol).to(DEV), ks).cpu().numpy()
        worst = max(worst, int((got != np.unpackbits(p[f'dil_{ks}'])[:vol.size].reshape(vol.shape)).sum()))
    return result('dilate_volume (bit-exact vs oracle C and reference golden)', worst, 0)


def check_isolate_tumor():
    from rsuper_amd.training import losses_foundation as lf
    p = golden('primitives')
    x = T(p['iso_x'])
    sh = tuple(x.shape)
    bad, notes = 0, []
    for name, dia, vol in [('a', 7.0, 150.0), ('b', 4.6, 40.0), ('c', 9.0, 300.0)]:
        m, ms, mb = lf.isolate_tumor(x.to(DEV), dia, True, 1.5, vol, 0.2, 0.2)
        for got, key in ((m, 'm'), (ms, 's'), (mb, 'b')):
            ref = np.unpackbits(p[f'iso_{name}_{key}'])[:x.numel()].reshape(sh)
            d = int((got.cpu().numpy() != ref).sum())
            bad += d
            notes.append(f'{name}{key}:{d}')
    xb = T(p['iso_border_x'])
    m, ms, mb = lf.isolate_tumor(xb.to(DEV), 9.0, True, 1.5, 380.0, 0.2, 0.2)
    for got, key in ((m, 'm'), (ms, 's'), (mb, 'b')):
        ref = np.unpackbits(p[f'iso_border_{key}'])[:xb.numel()].reshape(tuple(xb.shape))
        d = int((got.cpu().numpy() != ref).sum())
        bad += d
        notes.append(f'border{key}:{d}')
    # the speculative, device-resident search (no host reads between argmax, ball, top-k): where its two assumptions hold it must give the same
    # bits, where they do not (the border case needs the growth loop / dilation rounds) its check must say so -- the caller then repeats exactly
    spec_valid = 0
    for name, xx, dia, vol in [('a', x, 7.0, 150.0), ('b', x, 4.6, 40.0), ('c', x, 9.0, 300.0), ('border', xb, 9.0, 380.0)]:
        checks = []
        r3 = lf.isolate_tumor_spec(xx.to(DEV), dia, 1.5, vol, checks, 0.2, 0.2)
        ok, _ = lf._spec_ok(checks)
        ref3 = lf.isolate_tumor(xx.to(DEV), dia, True, 1.5, vol, 0.2, 0.2)
        if ok:
            spec_valid += 1
            d = sum(int((a != b).sum()) for a, b in zip(r3, ref3))
            bad += d
            notes.append(f'spec_{name}:{d}')
        else:
            notes.append(f'spec_{name}:fallback')
    if spec_valid == 0:
        bad += 1
        notes.append('speculation never valid')
    return result('isolate_tumor masks (bit-exact vs reference golden; speculative search == exact search)', bad, 0, ' '.join(notes))


def check_gwrp():
    from rsuper_amd.training import losses_foundation as lf
    p = golden('primitives')
    pm = T(p['gwrp_pm'])
    xsig = T(p['gwrp_x']).clamp(1e-6, 1 - 1e-6)
    logit = torch.log(xsig / (1 - xsig))
    w, n = lf.gwrp_foreground_weights(logit.to(DEV), pm.to(torch.uint8).to(DEV))
    ref = T(p['gwrp_w']) * pm.sum() * pm
    e = relerr(w.cpu(), ref)
    # the sort-based path of large pseudo masks: bit-identical to the pairwise-count kernel, on the fixture and on a 40 k-voxel mask with many ties
    old = lf.GWRP_SORT_ABOVE
    try:
        lf.GWRP_SORT_ABOVE = 0
        w2, _ = lf.gwrp_foreground_weights(logit.to(DEV), pm.to(torch.uint8).to(DEV))
        g = torch.Generator().manual_seed(5)
        big_x = (torch.randint(0, 50, (40, 40, 40), generator=g).float() / 7.0).to(DEV)        # few distinct values: thousands of ties
        big_pm = (torch.rand((40, 40, 40), generator=g) < 0.65).to(torch.uint8).to(DEV)
        wb_sort, nb = lf.gwrp_foreground_weights(big_x, big_pm)
        lf.GWRP_SORT_ABOVE = 1 << 30
        wb_pair, _ = lf.gwrp_foreground_weights(big_x, big_pm)
    finally:
        lf.GWRP_SORT_ABOVE = old
    same = bool(torch.equal(w2, w)) and bool(torch.equal(wb_sort, wb_pair))
    return result('gwrp_weights', e if same else float('inf'), 1e-4, f'N {n}; sort path == pairwise path on the fixture and on {nb} voxels with ties')


def make_args(**kw):
    d = dict(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1,
             volume_loss_tolerance=0.2, ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2,
             multi_ch_tumor=False, stardard_ce_ball=False, classification_branch=False, ema=True, ema_alpha=0.99)
    d.update(kw)
    return argparse.Namespace(**d)


LOSS_CASES = [('single_last', dict(loss='ball_dice_last'), False, 7, None),
              ('single_both', dict(loss='ball_dice_both'), False, 7, None),
              ('single_dice', dict(loss='dice'), False, 7, None),
              ('single_ball', dict(loss='ball'), False, 7, None),
              ('single_norep', dict(report_volume_loss_basic=0.0), False, 7, None),
              ('deep_last', dict(loss='ball_dice_last'), True, 7, None),
              ('deep_dice', dict(loss='dice'), True, 7, None),
              ('single_both_cw', dict(loss='ball_dice_both'), False, 7, 'cw'),
              ('single_both_norpt', dict(loss='ball_dice_both'), False, 8, None),
              ('multi_ch_both', dict(loss='ball_dice_both'), False, 'multi', None),      # lesion group over two channels (max-merge)
              ('multi_ch_deep_last', dict(loss='ball_dice_last'), True, 'multi', None)]


def loss_inputs(seed):
    if seed == 'multi':
        classes = synth.MULTI_CH_CLASSES
        bt = synth.multi_ch_batch(2, 32, ['mask', 'report'], seed=7, diam_range=(5.0, 9.0), max_tumors=2)
        return classes, bt, synth.logits(2, len(classes), 32, seed=199), synth.logits(2, len(classes), 32, seed=200)
    classes = synth.TINY_CLASSES
    kinds = ['mask', 'report'] if seed == 7 else ['healthy', 'mask']
    kw = dict(diam_range=(5.0, 9.0), max_tumors=2) if seed == 7 else {}
    bt = synth.batch(2, 32, classes, kinds, seed=seed, **kw)
    return classes, bt, synth.logits(2, len(classes), 32, seed=99), synth.logits(2, len(classes), 32, seed=100)


def check_calculate_loss(tag, akw, deep, seed, cw):
    from rsuper_amd.training import losses_foundation as lf
    g = golden('calc_loss')
    classes, bt, lg0, lg1 = loss_inputs(seed)
    a, b = T(lg0).to(DEV).requires_grad_(True), T(lg1).to(DEV).requires_grad_(True)
    res = lf.calculate_loss({'segmentation': [a, b] if deep else a}, T(bt['label']).to(DEV), T(bt['unk_channels']).to(DEV), make_args(**akw),
                            None, T(bt['mask']).to(DEV), T(bt['volumes']).to(DEV), T(bt['diameters']).to(DEV), classes,
                            class_weights=None if cw is None else T(g['cw']).to(DEV))
    res['overall'].backward()
    torch.cuda.synchronize()
    notes, worst = [], 0.0
    if sorted(res.keys()) != list(g[f'{tag}_keys']):
        return result(f'calculate_loss[{tag}]', float('inf'), 1e-4, f'keys {sorted(res.keys())} vs {list(g[tag + "_keys"])}')
    for k, v in res.items():
        d = abs(float(v.detach()) - float(g[f'{tag}_{k}']))
        worst = max(worst, d)
        notes.append(f'{k}:{d:.1e}')
    sub, _ = synth.subsample(a.grad.cpu().numpy(), 8192)
    ref = g[f'{tag}_g0_sub']
    eg = float(np.abs(sub - ref).max() / max(np.abs(ref).max(), 1e-12))
    notes.append(f'grad:{eg:.1e}')
    return result(f'calculate_loss[{tag}]', max(worst, eg), 1e-4, ' '.join(notes))


def check_isolate_tumor_large(name):
    """tests/golden/ball_large.npz part A (reference fixtures at d = 15 ... 40): the default path (separable two-stage correlation, device
    radix select) must give the reference's masks bit for bit; the speculative search must either agree or report that it cannot decide."""
    from rsuper_amd.training import losses_foundation as lf
    g = golden('ball_large')
    x, d, vol = synth.ball_case(name)
    sh = x.shape
    xt = T(x).to(DEV)
    got3 = lf.isolate_tumor(xt, d, True, 1.5, vol, 0.2, 0.2)
    tie = bool(int(g[f'iso_{name}_tie_dependent'][0]))
    pos = x > 0
    bad, notes = 0, []
    for got, key in zip(got3, 'msb'):
        ref = np.unpackbits(g[f'iso_{name}_{key}'])[:x.size].reshape(sh)
        gn = got.cpu().numpy()
        dd = int((gn[pos] != ref[pos]).sum()) if tie else int((gn != ref).sum())
        bad += dd
        notes.append(f'{key}:{dd}')
    checks = []
    r3 = lf.isolate_tumor_spec(xt, d, 1.5, vol, checks, 0.2, 0.2)
    ok, _ = lf._spec_ok(checks)
    loops = [int(v) for v in g[f'iso_{name}_loops']]
    if ok:
        dd = sum(int((a != b).sum()) for a, b in zip(r3, got3))
        bad += dd + (1 if any(loops) else 0)          # the reference looped -> the speculation must not have claimed validity
        notes.append(f'spec:{dd}')
    else:
        bad += 0 if any(loops) else 1                 # ... and the other way round
        notes.append('spec:fallback')
    return result(f'isolate_tumor[{name}] d={d:.0f} edge {sh[0]} (bit-exact vs reference golden' + (', positive voxels only: tie-dependent case)' if tie else ')'),
                  bad, 0, ' '.join(notes) + f' reference loops {loops}')


def check_calculate_loss_large(tag):
    """tests/golden/ball_large.npz part B: calculate_loss with three tumours of d = 15 ... 40 per report sample at 48^3 / 64^3."""
    from rsuper_amd.training import losses_foundation as lf
    g = golden('ball_large')
    classes, bt, lg0, lg1, loss, deep = synth.ball_loss_case_inputs(tag)
    a, b = T(lg0).to(DEV).requires_grad_(True), T(lg1).to(DEV).requires_grad_(True)
    res = lf.calculate_loss({'segmentation': [a, b] if deep else a}, T(bt['label']).to(DEV), T(bt['unk_channels']).to(DEV), make_args(loss=loss),
                            None, T(bt['mask']).to(DEV), T(bt['volumes']).to(DEV), T(bt['diameters']).to(DEV), classes)
    res['overall'].backward()
    torch.cuda.synchronize()
    if sorted(res.keys()) != list(g[f'{tag}_keys']):
        return result(f'calculate_loss_large[{tag}]', float('inf'), 1e-4, f'keys {sorted(res.keys())} vs {list(g[tag + "_keys"])}')
    notes, worst = [], 0.0
    for k, v in res.items():
        dd = abs(float(v.detach()) - float(g[f'{tag}_{k}']))
        worst = max(worst, dd)
        notes.append(f'{k}:{dd:.1e}')
    for t_, key in ((a, 'g0'),) + (((b, 'g1'),) if deep else ()):
        sub, _ = synth.subsample(t_.grad.cpu().numpy(), 8192)
        ref = g[f'{tag}_{key}_sub']
        eg = float(np.abs(sub - ref).max() / max(np.abs(ref).max(), 1e-12))
        worst = max(worst, eg)
        notes.append(f'{key}:{eg:.1e}')
    return result(f'calculate_loss_large[{tag}]', worst, 1e-4, ' '.join(notes))


def check_optimizer():
    from rsuper_amd.training.utils import FusedAdamWEMA, clip_grad_norm_
    g = synth.rng(9)
    shapes = [(33,), (8, 4, 3, 3, 3), (1, 7), (100000,)]
    ps = [T(g.standard_normal(s).astype(np.float32)) for s in shapes]
    gs = [[T(g.standard_normal(s).astype(np.float32) * 0.7) for s in shapes] for _ in range(3)]
    ref_p = [p.clone() for p in ps]
    ref_e = [p.clone() for p in ps]
    opt_ref = to.AdamW(ref_p, lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    dp = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    de = [p.clone().to(DEV) for p in ps]
    opt = FusedAdamWEMA(dp, lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    worst = 0.0
    for step in range(3):
        gr = [x.clone() for x in gs[step]]
        n_ref = to.clip_grad_norm_(gr, 1.0)
        opt_ref.step(gr)
        to.update_ema(ref_p, ref_e, 0.99, step)
        for p, gg in zip(dp, gs[step]):
            p.grad = gg.clone().to(DEV)
        n = opt.fused_step(max_norm=1.0, ema_params=de, ema_alpha=min(1 - 1 / (step + 1), 0.99))
        torch.cuda.synchronize()
        worst = max(worst, abs(float(n) - float(n_ref)) / float(n_ref))
        for a, b in zip(dp, ref_p):
            worst = max(worst, (a.detach().cpu() - b).abs().max().item())
        for a, b in zip(de, ref_e):
            worst = max(worst, (a.cpu() - b).abs().max().item())
    # stand-alone clip
    for p, gg in zip(dp, gs[0]):
        p.grad = gg.clone().to(DEV)
    clip_grad_norm_(dp, 1.0)
    gr = [x.clone() for x in gs[0]]
    to.clip_grad_norm_(gr, 1.0)
    worst = max(worst, max((a.grad.cpu() - b).abs().max().item() for a, b in zip(dp, gr)))
    return result('fused clip+AdamW+EMA', worst, 2e-6)


def check_train_steps(mode='f32'):
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    g = golden('train_step')
    classes, bt, _, _ = loss_inputs(7)
    net = make_tiny_unet(mode)
    ema = make_ema(net)
    opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    args = make_args(loss='ball_dice_both')
    batch = dict(image=T(synth.image(2, 32, seed=4321)).to(DEV), label=T(bt['label']).to(DEV), unk_channels=T(bt['unk_channels']).to(DEV),
                 mask=T(bt['mask']).to(DEV), volumes=T(bt['volumes']).to(DEV), diameters=T(bt['diameters']).to(DEV))
    notes, worst = [], 0.0
    tol = 1e-4 if mode == 'f32' else 3e-2
    for step in range(2):
        la, gn = train_step(net, ema, opt, batch, args, classes, step)
        torch.cuda.synchronize()
        for k, v in la.items():
            d = abs(float(v.detach()) - float(g[f's{step}_{k}']))
            worst = max(worst, d)
            notes.append(f's{step}.{k}:{d:.1e}')
        dgn = abs(float(gn) - float(g[f's{step}_gradnorm'])) / float(g[f's{step}_gradnorm'])
        notes.append(f's{step}.gnorm:{dgn:.1e}')
        worst = max(worst, dgn * tol / (5e-3 if mode == 'f32' else 0.2))
        for k in ['inc.conv1.weight', 'outc.weight', 'outc.bias']:
            p = dict(net.named_parameters())[k].detach().cpu().numpy().reshape(-1)[:64]
            d = np.abs(p - g[f's{step}_p_{k}_head'])
            notes.append(f's{step}.{k}:med{np.median(d):.1e}/max{d.max():.1e}')
            if mode == 'f32':
                worst = max(worst, (np.median(d) / 2e-5) * tol, (d.max() / 1.2e-3) * tol)
    return result(f'train_steps_golden[{mode}]', worst, tol, ' '.join(notes))


# ================================================================================================ registry
def with_strided(force, fn, *a):
    """Run a check with the strided convolutions forced onto the parity-class kernel (force '1', csrc/conv3d_igemm_s2.hip) or onto the
    stride-1 evaluation at full resolution (force '0'); the parity-class kernel is the default (ops.strided_kernel)."""
    old = os.environ.get('RSUPER_S2_KERNEL')
    os.environ['RSUPER_S2_KERNEL'] = force
    try:
        r = fn(*a)
    finally:
        if old is None:
            del os.environ['RSUPER_S2_KERNEL']
        else:
            os.environ['RSUPER_S2_KERNEL'] = old
    r['name'] = f"s2kernel={force}:{r['name']}"
    return r


def with_variant(variant, fn, *a):
    """Run a conv check under a forced igemm kernel variant (0 classic, 1 producer/consumer, 2 = round-1 auto choice,
    4 = weight-stationary kernel on every 32-column launch, 6 / 7 = volume-fitted K-split kernel with the box
    chosen per volume / the 4x4x4 box, 8 = depth-reuse kernel for every launch with more than 32 columns); restores the default (3)."""
    from rsuper_amd.hip import ops
    L = ops._L()
    L.rsuper_conv3_variant(variant)
    try:
        r = fn(*a)
    finally:
        L.rsuper_conv3_variant(3)
    r['name'] = f"v{variant}:{r['name']}"
    return r


def with_wgrad2(fn, *a):
    """Run a check with EVERY supported bf16 weight gradient on the second-generation kernel (csrc/conv3d_wgrad2.hip), whatever the number of tiles
    per block (the default takes it from 12 tiles per block: the small test volumes would never reach it)."""
    from rsuper_amd.hip import ops
    L = ops._L()
    old = L.rsuper_conv3_wgrad2_min_tiles(-1)
    L.rsuper_conv3_wgrad2_min_tiles(0)
    try:
        r = fn(*a)
    finally:
        L.rsuper_conv3_wgrad2_min_tiles(old)
    r['name'] = f"wgrad2:{r['name']}"
    return r


def with_wgrad_classic(fn, *a):
    """Run a check with the round-1..3 tile-streaming weight gradient (csrc/conv3d_wgrad.hip) for every bf16 launch: neither the second-generation nor the
    small-volume kernel (the default takes one of the two for most shapes)."""
    from rsuper_amd.hip import ops
    L = ops._L()
    old = L.rsuper_conv3_wgrad2_min_tiles(-1)
    L.rsuper_conv3_wgrad2_min_tiles(1 << 20)
    try:
        r = fn(*a)
    finally:
        L.rsuper_conv3_wgrad2_min_tiles(old)
    r['name'] = f"wgrad-classic:{r['name']}"
    return r


def check_wgrad_sliced_dy(N, S, Ca, Cout, seed=0):
    """Weight gradient with the two dY sources given as channel slices of ONE tensor (ld = 2 Cout, the layout the zero-stuffed evaluation of the strided
    block and BasicBlockFn's [dY1 | dOut] use) and a normalised x source, against the float64 gradient on the bf16 operands the kernel sees."""
    from rsuper_amd.hip import ops
    D, H, W = S
    dt = torch.bfloat16
    xa = rnd(_rng_t(seed + 1, (N, Ca, D, H, W)) + 0.3, 'bf16')
    mr = stats_ref(xa)
    xh = torch.relu((xa.double() - mr[:, :, 0].double()[:, :, None, None, None]) * mr[:, :, 1].double()[:, :, None, None, None]).float().bfloat16().double()
    dy = _rng_t(seed + 6, (N, 2 * Cout, D, H, W)).bfloat16()
    refs = []
    for part in (dy[:, :Cout], dy[:, Cout:]):
        w = torch.zeros((Cout, Ca, 3, 3, 3), dtype=torch.float64, requires_grad=True)
        F.conv3d(xh, w, padding=1).backward(part.double())
        refs.append(w.grad.float())
    dfull = to_cl(dy.float(), dt)
    dw1 = torch.full((Cout, Ca, 3, 3, 3), float('nan'), device=DEV)
    dw2 = torch.full((Cout, Ca, 3, 3, 3), float('nan'), device=DEV)
    ops.wgrad(ops.Src(to_cl(xa, dt), mr=mr.to(DEV)), None, ops.Src(dfull, C=Cout), ops.Src(dfull, C=Cout, off=Cout), dw1, dw2, (N, D, H, W))
    torch.cuda.synchronize()
    e = max(relerr(dw1.cpu(), refs[0]), relerr(dw2.cpu(), refs[1]))
    return result(f'wgrad_sliced_dy[bf16 N{N} S{S} {Ca}->2x{Cout}]', e, 3e-3, f'dw {e:.2e}')


def all_checks(quick=False):
    cs = []
    for mode in ('f32', 'bf16'):
        cs += [
            (check_conv_fwd, (mode, 1, (4, 4, 16), 32, 0, 32)),
            (check_conv_fwd, (mode, 2, (8, 8, 16), 32, 0, 32, False, True)),
            (check_conv_fwd, (mode, 1, (5, 6, 7), 8, 16, 8, True, False)),
            (check_conv_fwd, (mode, 1, (8, 12, 20), 64, 32, 64, True, False)),
            (check_conv_fwd, (mode, 1, (4, 8, 16), 64, 0, 64, False, False, False)),
            (check_conv_fwd, (mode, 1, (6, 6, 6), 80, 0, 80, False, True)),
            (check_conv_bwd, (mode, 1, (4, 4, 16), 32, 0, 32, False)),
            (check_conv_bwd, (mode, 2, (8, 8, 16), 32, 0, 32, False)),
            (check_conv_bwd, (mode, 1, (5, 6, 7), 8, 16, 8, True)),
            (check_conv_bwd, (mode, 1, (8, 12, 20), 64, 32, 64, True)),
            (check_conv_bwd, (mode, 1, (6, 6, 6), 80, 0, 80, False)),
            (check_pool, (mode,)), (check_pool_skip, (mode,)), (check_pool_skip, (mode, (1, 7, 9, 5, 8), 65)), (check_pool_skip, (mode, (2, 24, 24, 24, 32), 66)),
            (check_upsample, (mode,)), (check_upsample, (mode, 16, 6, 12, 70)), (check_upsample, (mode, 8, 8, 32, 71)), (check_upsample, (mode, 8, 2, 4, 72)),
            (check_upsample, (mode, 8, 5, 9, 73)),
            (check_stem, (mode,)), (check_stem, (mode, 32, 16)), (check_head, (mode,)), (check_head, (mode, 32, 42, 12)),
            (check_stem, (mode, 64, 12)), (check_head, (mode, 64, 70, 10)), (check_head, (mode, 16, 130, 9)), (check_unet_wide, (mode,)),
            (check_basic_block, (mode, 'b8_16', 8, 16, 12, 1)), (check_basic_block, (mode, 'b16_16', 16, 16, 10, 2)),
            (check_basic_block, (mode, 'b24_8', 24, 8, 12, 3)),
            (check_basic_block, (mode, 'b8_16_s2', 8, 16, 12, 4)), (check_basic_block, (mode, 'b16_16_s2', 16, 16, 9, 5)),
            (check_unet_tiny_nopool, (mode,)),
            (check_unet_tiny, (mode,)),
        ]
        for force in ('0',):             # the rounds-1/2 evaluation of the strided convolutions stays selectable (RSUPER_S2_KERNEL=0): keep it pinned too
            cs += [(with_strided, (force, check_basic_block, mode, 'b8_16_s2', 8, 16, 12, 4)), (with_strided, (force, check_basic_block, mode, 'b16_16_s2', 16, 16, 9, 5)),
                   (with_strided, (force, check_unet_tiny_nopool, mode))]
    cs += [(check_token_attn, (2, 81, 10, 32)), (check_token_attn, (1, 128, 2, 32)), (check_token_attn, (3, 7, 3, 16)), (check_token_attn, (2, 65, 4, 24)),
           (check_token_attn, (1, 1, 1, 4)), (check_token_attn, (2, 96, 2, 64))]      # fusion transformer's attention core: shipped shape, limits, ragged
    for m in ('f32', 'bf16'):           # small volumes (the 6^3 level as shipped, ragged shapes, more samples than tiles): tiles mostly empty, fitted igemm kernels, split-K epilogue
        cs += [(check_conv_bwd, (m, 2, (6, 6, 6), 64, 0, 64, False)), (check_conv_bwd, (m, 2, (6, 6, 6), 32, 0, 48, True)), (check_conv_bwd, (m, 3, (5, 6, 4), 40, 0, 24, True)),
               (check_conv_bwd, (m, 1, (2, 2, 2), 8, 0, 8, False)), (check_conv_bwd, (m, 9, (3, 2, 7), 16, 0, 16, False)), (check_conv_bwd, (m, 2, (6, 6, 6), 320, 0, 320, False))]
    # the split shape of the <= 6^3 box kernel needs nsplit x N x 216 x n_cols floats of workspace; beyond N x ceil(n_cols / 32) = 256
    # that exceeds what is registered -> these batches must take another shape instead of writing past it (N = 13 / 640 columns is the first such case)
    cs += [(check_conv_bwd, ('bf16', 32, (6, 6, 6), 320, 0, 320, False)), (check_conv_bwd, ('bf16', 32, (6, 6, 6), 320, 0, 320, True)),
           (check_conv_bwd, ('bf16', 13, (6, 6, 6), 64, 0, 320, True)), (check_conv_fwd, ('bf16', 26, (6, 6, 6), 64, 0, 320, False, True))]
    # pre-normalised sources (LDS-DMA fed weight gradient): one / two x sources, one / two dY sources, 32- and 64-row blocks, ragged volumes (zero padding
    # by out-of-range DMA lanes on every face), channel tails, several tiles per block, both samples in one block's range, >= 128 tiles (XCD-aware order)
    cs += [(check_wgrad_xhat, (1, (4, 4, 16), 32, 0, 32, 0)), (check_wgrad_xhat, (2, (8, 8, 32), 32, 0, 32, 0, 1)), (check_wgrad_xhat, (1, (5, 6, 7), 8, 16, 8, 8, 2)),
           (check_wgrad_xhat, (2, (8, 12, 20), 64, 32, 64, 64, 3)), (check_wgrad_xhat, (1, (7, 9, 35), 40, 0, 24, 0, 4)), (check_wgrad_xhat, (2, (16, 16, 64), 32, 32, 32, 32, 5)),
           (check_wgrad_xhat, (2, (24, 24, 48), 32, 0, 32, 0, 6)), (check_wgrad_xhat, (1, (16, 32, 64), 64, 0, 64, 0, 7)), (check_wgrad_xhat, (2, (12, 12, 12), 128, 0, 128, 0, 8)),
           (check_wgrad_xhat, (3, (6, 6, 6), 32, 0, 96, 32, 9)), (check_wgrad_xhat, (1, (2, 3, 5), 8, 0, 8, 0, 10))]
    # the second-generation weight gradient on every bf16 backward case and on the pre-normalised ones (the default dispatch needs >= 12 tiles per block)
    cs += [(with_wgrad2, (fn,) + a) for fn, a in list(cs) if fn in (check_conv_bwd, check_wgrad_xhat) and (fn is check_wgrad_xhat or a[0] == 'bf16')]
    cs += [(with_wgrad_classic, (fn,) + a) for fn, a in list(cs) if fn is check_conv_bwd and a[0] == 'bf16']
    # small volumes as the UNet has them (12^3 with 2 depth parts, 6^3 whole, ragged, two x / dY sources, channel tails): csrc/conv3d_wgrad_sv.hip by default
    cs += [(check_conv_bwd, ('bf16', 2, (12, 12, 12), 64, 0, 64, False)), (check_conv_bwd, ('bf16', 2, (12, 12, 12), 32, 64, 32, True)),
           (check_conv_bwd, ('bf16', 1, (11, 9, 13), 40, 0, 24, True)), (check_wgrad_sliced_dy, (2, (12, 12, 12), 64, 64, 6)),
           (check_wgrad_xhat, (2, (12, 12, 12), 96, 32, 64, 32, 12)), (check_wgrad_xhat, (4, (6, 6, 6), 64, 0, 96, 0, 13)),
           # 64 rows at 576 tiles: the second-generation kernel as two 32-row blocks per chunk (rs_wgrad2_mt1), one and two dY sources
           (check_wgrad_xhat, (2, (32, 48, 48), 64, 0, 64, 0, 14)), (check_wgrad_sliced_dy, (2, (32, 48, 48), 32, 32, 7))]
    cs += [(with_wgrad2, (check_conv_bwd, 'bf16', 2, (16, 16, 64), 32, 32, 64, True)), (with_wgrad2, (check_conv_bwd, 'bf16', 1, (16, 32, 64), 64, 0, 64, False)),
           (with_wgrad2, (check_conv_bwd, 'bf16', 2, (24, 24, 48), 32, 0, 32, False)),
           (check_wgrad_sliced_dy, (1, (23, 23, 23), 64, 128)), (with_wgrad2, (check_wgrad_sliced_dy, 1, (23, 23, 23), 64, 128)),
           (with_wgrad2, (check_wgrad_sliced_dy, 2, (9, 10, 19), 32, 32, 1)), (check_wgrad_sliced_dy, (1, (47, 47, 47), 64, 128, 2)),
           # last w tile with its first invalid voxel at positions 8 .. 15 (W mod 16 >= 8: the validity bits of the upper half of a tile row), several tiles per block
           (with_wgrad2, (check_wgrad_sliced_dy, 1, (9, 9, 31), 64, 128, 3)), (with_wgrad2, (check_wgrad_sliced_dy, 2, (6, 7, 25), 32, 32, 4)),
           (with_wgrad2, (check_wgrad_sliced_dy, 2, (10, 12, 47), 128, 128, 5)), (with_wgrad2, (check_wgrad_xhat, 2, (11, 9, 29), 64, 64, 64, 64, 11))]
    for m in ('f32', 'bf16'):           # strided weight gradient: even / odd / ragged sizes, one and two dy sources, channel tails, several tiles per split
        cs += [(check_wgrad_s2, (m, 1, (8, 8, 32), 32, 32, 0)), (check_wgrad_s2, (m, 2, (12, 10, 20), 16, 32, 32)), (check_wgrad_s2, (m, 1, (7, 9, 35), 8, 16, 16)),
               (check_wgrad_s2, (m, 2, (5, 17, 66), 40, 24, 24)), (check_wgrad_s2, (m, 1, (2, 3, 5), 8, 8, 8)), (check_wgrad_s2, (m, 3, (24, 24, 24), 64, 128, 128))]
    # trilinear backward, row-sharing kernel: the UNet's shapes (scaled down), odd / ragged sizes, scale 1 and ~3, many channels
    cs += [(check_upsample_bwd, (2, (6, 6, 6), (12, 12, 12), 64)), (check_upsample_bwd, (1, (12, 12, 12), (24, 24, 24), 16, 81)),
           (check_upsample_bwd, (2, (3, 5, 7), (6, 9, 13), 8, 82)), (check_upsample_bwd, (1, (5, 4, 3), (5, 4, 3), 8, 83)),
           (check_upsample_bwd, (1, (3, 3, 4), (8, 9, 10), 24, 84)), (check_upsample_bwd, (1, (1, 2, 3), (2, 4, 6), 8, 85)),
           (check_upsample_bwd, (2, (24, 24, 24), (48, 48, 48), 16, 86)), (check_upsample_bwd, (1, (6, 6, 6), (12, 12, 12), 320, 87))]
    for m in ('f32', 'bf16'):           # 1x1x1 convolutions of MedFormer's attention stages on MFMA: shipped shapes, ragged rows / channels
        cs += [(check_pointwise, (m, 27648, 128, 512, False)), (check_pointwise, (m, 13824, 512, 128, True)), (check_pointwise, (m, 3456, 256, 1024, False)),
               (check_pointwise, (m, 3456, 1024, 256, True)), (check_pointwise, (m, 432, 320, 1280, False)),       # reduction split over the waves
               (check_pointwise, (m, 100, 36, 20, True)), (check_pointwise, (m, 33, 4, 4, False)), (check_pointwise, (m, 1000, 72, 260, True))]
    for variant in (0, 1, 4, 6, 7, 8):  # every bf16 igemm kernel on every conv case (8: the depth-reuse kernel wherever its limits allow) (6 / 7: the volume-fitted K-split kernel, incl. its split-reduction shape) (the default picks per launch)
        cs += [(with_variant, (variant, fn) + a) for fn, a in list(cs) if fn in (check_conv_fwd, check_conv_bwd) and a[0] == 'bf16']
    # bf16 forward / data gradient against float64 on the same bf16 operands, one-rounding bound: ragged volumes (every face of the halo),
    # one / two sources, fused shortcut (128 / 96 / 64 / 16 columns), residual, raw and normalised sources, the low-resolution box shapes; default dispatch
    # and every forced kernel variant
    exact = [('fwd', 1, (5, 9, 19), 32, 0, 32), ('fwd', 2, (8, 8, 16), 32, 0, 32, False, True), ('fwd', 1, (8, 12, 20), 64, 32, 64, True),
             ('fwd', 1, (4, 8, 16), 64, 0, 64, False, False, False), ('fwd', 1, (6, 6, 6), 80, 0, 80, False, True), ('fwd', 1, (5, 6, 7), 8, 16, 8, True),
             ('fwd', 1, (12, 20, 48), 64, 64, 64, True, False, True, 3), ('fwd', 2, (12, 12, 12), 128, 0, 128, False, True, True, 4),
             ('fwd', 2, (9, 17, 33), 32, 64, 32, True, False, True, 5),
             ('dgrad', 1, (5, 9, 19), 32, 0, 32), ('dgrad', 1, (8, 12, 20), 64, 32, 64, True), ('dgrad', 2, (6, 6, 6), 64, 0, 64),
             ('dgrad', 2, (12, 12, 12), 128, 0, 128, False, False, True, 6), ('dgrad', 1, (5, 6, 7), 8, 16, 8, True), ('dgrad', 2, (9, 17, 33), 32, 64, 32, True, False, True, 7),
             ('dgrad', 1, (12, 20, 48), 128, 64, 64, True, False, True, 8)]
    cs += [(check_conv_exact, a) for a in exact]
    cs += [(check_conv_mixed_sources, (2, (32, 64, 64), 32, 32, 64))]
    # stride 2 (row g): odd sizes 47 / 13, ragged, even, several bricks per persistent block, channel tails; both kernel generations
    for kern in ('1', '0'):
        cs += [(check_conv_exact_s2, (kind, N, S, Ca, Co, sd, kern)) for kind in ('fwd', 'dgrad') for N, S, Ca, Co, sd in
               [(1, (47, 47, 47), 16, 32, 1), (1, (13, 13, 13), 16, 32, 2), (2, (24, 24, 24), 32, 64, 3), (1, (9, 20, 35), 48, 24, 4), (2, (48, 48, 48), 32, 64, 5)]]
    # the shapes the depth-reuse kernel takes under the DEFAULT dispatch (>= 400 tiles of 4x8x16: 432 / 512 tiles on 256 persistent blocks, so a block's
    # statistics registers run over several tiles): 64 -> 64 @48^3 with residual, an up-block head 64 + 32 -> 32 + shortcut
    cs += [(check_conv_exact, ('fwd', 2, (48, 48, 48), 64, 0, 64, False, True, True, 9)), (check_conv_exact, ('fwd', 2, (32, 64, 64), 64, 32, 32, True, False, True, 10)),
           (check_conv_exact, ('dgrad', 2, (32, 64, 64), 64, 32, 32, True, False, True, 11))]
    for variant in (0, 1, 4, 6, 7, 8):
        cs += [(with_variant, (variant, check_conv_exact) + a) for a in exact]
    cs += [(with_variant, (1, check_conv_fwd, 'bf16', 2, (8, 24, 32), 32, 0, 32, False, True)),       # persistent: several tiles per block
           (with_variant, (1, check_conv_fwd, 'bf16', 1, (12, 20, 48), 64, 64, 128, True, False)),    # 128 columns, 4 chunks, 2 sources
           (with_variant, (1, check_conv_bwd, 'bf16', 2, (8, 24, 32), 64, 64, 64, True)),
           (with_variant, (1, check_conv_bwd, 'bf16', 1, (12, 20, 48), 128, 0, 128, False))]
    cs += [(with_variant, (4, check_conv_fwd, 'bf16', 2, (8, 24, 32), 32, 0, 32, False, True)),       # weight-stationary kernel: several tiles per block,
           (with_variant, (4, check_conv_fwd, 'bf16', 1, (12, 20, 48), 64, 64, 128, True, False)),    # 4 chunks / 2 sources (weights re-streamed), ragged tiles,
           (with_variant, (4, check_conv_fwd, 'bf16', 2, (16, 32, 64), 64, 0, 64, False, True)),      # 64-column blocks with residual
           (with_variant, (4, check_conv_bwd, 'bf16', 2, (8, 24, 32), 64, 64, 64, True)),
           (with_variant, (4, check_conv_bwd, 'bf16', 1, (12, 20, 48), 128, 0, 128, False)),
           (with_variant, (4, check_conv_bwd, 'bf16', 2, (16, 32, 64), 32, 64, 32, True))]            # 96-column data gradient (three 32-column tiles)]
    cs += [(check_conv_bwd, ('bf16', 2, (16, 16, 64), 32, 32, 64, True)),      # >= 128 tiles, M = 128: 27-tap / 8-wave weight-gradient config
           (check_conv_bwd, ('bf16', 1, (16, 32, 64), 64, 0, 64, False))]
    cs += [(check_conv_bwd, ('bf16', 1, (4, 4, 16), 32, 0, 32, False, 0, 0)), (check_conv_bwd, ('bf16', 1, (8, 12, 20), 64, 32, 64, True, 0, 0)),
           (check_tr16_probe, ()),
           (check_plane_partials, ()), (check_seg_from_sums, (2, 26, False, 1)), (check_seg_from_sums, (3, 5, True, 2)),
           (check_seg_from_sums, (1, 300, True, 3)), (check_dilate, ()), (check_isolate_tumor, ()), (check_gwrp, ()), (check_optimizer, ())]
    cs += [(check_calculate_loss, c) for c in LOSS_CASES]
    cs += [(check_isolate_tumor_large, (n,)) for n in synth.BALL_CASES] + [(check_calculate_loss_large, (t_,)) for t_ in synth.BALL_LOSS_CASES]
    cs += [(check_train_steps, ('f32',)), (check_train_steps, ('bf16',))]
    cs += [(check_cnorm, (8, (6, 7, 9), True)), (check_cnorm, (72, (5, 4, 11), False, 1)), (check_cnorm, (1280, (3, 3, 3), True)),
           (check_cnorm, (256, (24, 24, 24), True, 1))]
    cs += [(check_linear_splitk, (27648, 128, 512, False)), (check_linear_splitk, (8192, 72, 24, True)), (check_linear_splitk, (10000, 16, 8, True))]
    cs += [(check_cl_planar, (2, (24, 24, 24), 32, 26)), (check_cl_planar, (1, (5, 7, 9), 8, 3)), (check_cl_planar, (2, (4, 4, 33), 64, 64))]
    cs += [(check_squeeze_excite, (8, (6, 7, 9))), (check_squeeze_excite, (1024, (12, 12, 12))), (check_squeeze_excite, (72, (5, 4, 3), 1))]
    cs += [(check_battn, (2, 1728, 27, 8, 32)), (check_battn, (1, 216, 27, 10, 32)), (check_battn, (2, 13824, 27, 4, 32)),
           (check_battn, (2, 100, 27, 1, 32)), (check_battn, (1, 512, 8, 2, 16)), (check_battn, (2, 61, 8, 5, 16)), (check_battn, (1, 8, 8, 4, 16))]
    cs += [(check_depthwise, (36, (6, 32, 40), 1)), (check_depthwise, (8, (7, 33, 35))), (check_depthwise, (64, (13, 48, 32), 1))]     # LDS-tiled kernel (planes >= 32 x 32)
    cs += [(check_depthwise, (8, (6, 7, 9))), (check_depthwise, (72, (5, 4, 11), 1)), (check_depthwise, (256, (12, 12, 12))),
           (check_depthwise, (1280, (3, 3, 3)))]
    cs += [(check_medformer_tiny, ('f32',)), (check_medformer_tiny, ('bf16',))]
    return cs
