"""A/B of the persistent strided forward (csrc/conv3d_igemm_s2k.hip) against the parity-class kernel (csrc/conv3d_igemm_s2.hip) on the same operands:
output (bf16: identical up to roundings of different fp32 summation orders), statistics rows reduced per column, and timing.  Usage: python tools/check_s2k.py"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsuper_amd.hip import ops

dev, dt = 'cuda', torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def run(N, S, Ca, Cout, which):
    os.environ['RSUPER_S2K'] = which
    g = torch.Generator().manual_seed(3)
    xa = (torch.randn((N, S, S, S, Ca), generator=g) * 1.5 + 0.3).to(dev).to(dt)
    xf = xa.float()
    m = xf.mean(dim=(1, 2, 3)); v = xf.var(dim=(1, 2, 3), unbiased=False)
    mra = torch.stack([m, 1.0 / torch.sqrt(v + 1e-5)], -1).contiguous()
    w1 = (torch.randn((Cout, Ca, 3, 3, 3), generator=g) / math.sqrt(27 * Ca)).to(dev)
    ws = (torch.randn((Cout, Ca, 3, 3, 3), generator=g) / math.sqrt(27 * Ca)).to(dev)
    nc = 2 * Cout
    O = (S + 1) // 2
    wp = ops.pack_weights(dt, 0, w1, ws, Ca, 0, Cout, Cout, 64)
    ys = torch.full((N, O, O, O, nc), float('nan'), device=dev, dtype=dt)
    rows = ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 1, Ca, 0, nc, N, S, S, S)
    part = torch.full((N, rows, nc, 2), float('nan'), device=dev, dtype=torch.float32)
    sa = ops.Src(xa, mr=mra)
    ops.igemm_s2(1, sa, None, wp, nc, (N, S, S, S), ys, part)
    torch.cuda.synchronize()
    t = timeit(lambda: ops.igemm_s2(1, sa, None, wp, nc, (N, S, S, S), ys, part))
    # float64 reference on the same bf16 operands
    ref = None
    if S <= 48:
        xh = torch.relu((xf - m[:, None, None, None, :]) * mra[..., 1][:, None, None, None, :]).to(dt).double().permute(0, 4, 1, 2, 3)
        wcat = torch.cat([w1, ws], 0).to(dt).double()
        ref = torch.nn.functional.conv3d(xh, wcat, stride=2, padding=1).permute(0, 2, 3, 4, 1)
    return ys.float(), part.sum(1), rows, t, ref


ok = True
for N, S, Ca, Cout in [(2, 96, 32, 64), (1, 47, 64, 128), (2, 48, 64, 128), (2, 24, 128, 256), (1, 13, 16, 32), (1, 20, 48, 24)]:
    ya, pa, ra, ta, ref = run(N, S, Ca, Cout, '1')
    yb, pb, rb, tb, _ = run(N, S, Ca, Cout, '0')
    scale = float(yb.abs().max())
    dy = float((ya - yb).abs().max()) / scale
    dp = float((pa - pb).abs().max() / pb.abs().max())
    gf = 2.0 * N * ((S + 1) // 2) ** 3 * 2 * Cout * Ca * 27 / 1e9
    line = f'N {N} S {S:3d} {Ca:3d} -> 2 x {Cout:3d}: rows {ra:4d} / {rb:4d}  new {ta:7.1f} us ({gf / ta * 1e3:6.1f} TF)  old {tb:7.1f} us ({gf / tb * 1e3:6.1f} TF)  out max diff {dy:.2e} of max  stats diff {dp:.2e}'
    if ref is not None:
        ea = float((ya.double() - ref).abs().max()) / scale
        eb = float((yb.double() - ref).abs().max()) / scale
        line += f'  |vs f64: new {ea:.2e} old {eb:.2e}'
        ok = ok and ea <= 1.5 * eb + 1e-6
    ok = ok and not math.isnan(dy) and dy < 1e-2 and dp < 1e-3
    print(line, flush=True)


def run_d(N, S, Cin, Cout, which):
    """data gradient: [dy1 | dOut] on the half grid -> dx on the full grid (masked by the forward input's ReLU), InstanceNorm-backward sums"""
    os.environ['RSUPER_S2D'] = which
    g = torch.Generator().manual_seed(5)
    O = (S + 1) // 2
    xa = (torch.randn((N, S, S, S, Cin), generator=g) * 1.5 + 0.3).to(dev).to(dt)
    xf = xa.float()
    m = xf.mean(dim=(1, 2, 3)); v = xf.var(dim=(1, 2, 3), unbiased=False)
    mra = torch.stack([m, 1.0 / torch.sqrt(v + 1e-5)], -1).contiguous()
    dy1 = torch.randn((N, O, O, O, Cout), generator=g).to(dev).to(dt); dy2 = torch.randn((N, O, O, O, Cout), generator=g).to(dev).to(dt)
    w1 = (torch.randn((Cout, Cin, 3, 3, 3), generator=g) / math.sqrt(27 * Cin)).to(dev)
    ws = (torch.randn((Cout, Cin, 3, 3, 3), generator=g) / math.sqrt(27 * Cin)).to(dev)
    wpd = ops.pack_weights(dt, 1, w1, ws, Cout, Cout, Cin, 0, 64)
    g0 = torch.full((N, S, S, S, Cin), float('nan'), device=dev, dtype=dt)
    rows = ops._L().rsuper_conv3_s2_part_rows(ops._DT[dt], 2, Cout, Cout, Cin, N, S, S, S)
    part = torch.full((N, rows, Cin, 2), float('nan'), device=dev, dtype=torch.float32)
    sa = ops.Src(xa, mr=mra)
    fn = lambda: ops.igemm_s2(2, ops.Src(dy1), ops.Src(dy2), wpd, Cin, (N, S, S, S), g0, part, ea=sa)
    fn()
    torch.cuda.synchronize()
    t = timeit(fn)
    ref = None
    if S <= 48:
        wcat = torch.cat([w1, ws], 0).to(dt).double()
        dy = torch.cat([dy1, dy2], -1).double().permute(0, 4, 1, 2, 3)
        dx = torch.nn.functional.conv_transpose3d(dy, wcat, stride=2, padding=1, output_padding=1 if S % 2 == 0 else 0)
        dx = dx[:, :, :S, :S, :S].permute(0, 2, 3, 4, 1)
        xn = (xf - m[:, None, None, None, :]) * mra[..., 1][:, None, None, None, :]
        ref = torch.where(xn > 0, dx, torch.zeros_like(dx))
    return g0.float(), part.sum(1), rows, t, ref


for N, S, Cin, Cout in [(2, 96, 32, 64), (1, 47, 64, 128), (2, 48, 64, 128), (2, 24, 128, 256), (1, 13, 16, 32), (1, 20, 48, 24)]:
    ya, pa, ra, ta, ref = run_d(N, S, Cin, Cout, '1')
    yb, pb, rb, tb, _ = run_d(N, S, Cin, Cout, '0')
    scale = float(yb.abs().max())
    dy = float((ya - yb).abs().max()) / scale
    dp = float((pa - pb).abs().max() / pb.abs().max())
    gf = 2.0 * N * ((S + 1) // 2) ** 3 * 2 * Cout * Cin * 27 / 1e9
    line = f'dgrad N {N} S {S:3d} 2 x {Cout:3d} -> {Cin:3d}: rows {ra:4d} / {rb:4d}  new {ta:7.1f} us ({gf / ta * 1e3:6.1f} TF)  old {tb:7.1f} us ({gf / tb * 1e3:6.1f} TF)  out max diff {dy:.2e} of max  sums diff {dp:.2e}'
    if ref is not None:
        ea = float((ya.double() - ref).abs().max()) / scale
        eb = float((yb.double() - ref).abs().max()) / scale
        line += f'  |vs f64: new {ea:.2e} old {eb:.2e}'
        ok = ok and ea <= 1.5 * eb + 1e-6
    ok = ok and not math.isnan(dy) and dy < 1e-2 and dp < 2e-3
    print(line, flush=True)
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
