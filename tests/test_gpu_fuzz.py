"""Seeded shape fuzzing of the convolution kernels against the torch fp32 reference (tests/gpu_checks.py): ragged spatial
sizes, channel counts that leave partially filled 32-channel chunks / 32-column tiles, one or two sources, with and without
the fused shortcut and residual -- under the default kernel choice and with each igemm variant forced."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import gpu_checks as gc  # noqa: E402

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        S = (int(rng.integers(3, 14)), int(rng.integers(3, 14)), int(rng.integers(5, 40)))
        Ca = int(rng.choice([8, 16, 24, 32, 40, 64, 72]))
        Cb = int(rng.choice([0, 0, 8, 32, 48]))
        Cout = int(rng.choice([8, 16, 24, 32, 40, 64, 96]))
        N = int(rng.integers(1, 3))
        sc = bool(rng.integers(0, 2)) or (Cb > 0) or (Ca != Cout)
        out.append((N, S, Ca, Cb, Cout, sc))
    return out


CASES = _cases(10, 2026)


@pytest.mark.parametrize('variant', [3, 4, 0, 1, 6, 7])
@pytest.mark.parametrize('case', CASES, ids=[f'N{c[0]}_S{"x".join(map(str, c[1]))}_{c[2]}+{c[3]}to{c[4]}_sc{int(c[5])}' for c in CASES])
def test_conv_fuzz_bf16(case, variant):
    N, S, Ca, Cb, Cout, sc = case
    r = gc.with_variant(variant, gc.check_conv_fwd, 'bf16', N, S, Ca, Cb, Cout, sc, False)
    assert r['ok'], (r['name'], r['err'], r['note'])
    r = gc.with_variant(variant, gc.check_conv_bwd, 'bf16', N, S, Ca, Cb, Cout, sc)
    assert r['ok'], (r['name'], r['err'], r['note'])


@pytest.mark.parametrize('case', CASES[:4], ids=[f'f32_{i}' for i in range(4)])
def test_conv_fuzz_f32(case):
    N, S, Ca, Cb, Cout, sc = case
    r = gc.check_conv_fwd('f32', N, S, Ca, Cb, Cout, sc, False)
    assert r['ok'], (r['name'], r['err'], r['note'])
    r = gc.check_conv_bwd('f32', N, S, Ca, Cb, Cout, sc)
    assert r['ok'], (r['name'], r['err'], r['note'])


def _glue_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        dims = (int(rng.integers(1, 10)), int(rng.integers(1, 10)), int(rng.integers(1, 23)))
        out.append((int(rng.choice([4, 8, 60, 64, 68, 132, 320])), dims, int(rng.integers(1, 3)), bool(rng.integers(0, 2))))
    return out


GLUE_CASES = _glue_cases(12, 77)


@pytest.mark.parametrize('case', GLUE_CASES, ids=[f'C{c[0]}_{"x".join(map(str, c[1]))}_N{c[2]}_r{int(c[3])}' for c in GLUE_CASES])
def test_attention_stage_kernels_fuzz(case):
    """Depthwise 3x3x3 (forward / data gradient / weight gradient) and stand-alone InstanceNorm [+ ReLU] (forward / backward) on ragged
    volumes: widths that are not multiples of the 4-voxel run, 1-voxel axes, channel counts that leave a partial 64-channel group, and
    both the one-launch (<= 512 voxels) and the three-launch InstanceNorm paths."""
    C, dims, N, relu = case
    r = gc.check_depthwise(C, dims, N)
    assert r['ok'], f"{r['name']}: err {r['err']:.3e} > tol {r['tol']:.1e}"
    if dims[0] * dims[1] * dims[2] > 1:                     # InstanceNorm of a single voxel is 0 / 0 in the reference too
        r = gc.check_cnorm(C, dims, relu, N)
        assert r['ok'], f"{r['name']}: err {r['err']:.3e} > tol {r['tol']:.1e}"


def _attn_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        T, dh = ((27, 32), (8, 16))[int(rng.integers(0, 2))]
        out.append((int(rng.integers(1, 4)), int(rng.choice([1, 7, 63, 64, 65, 200, 513, 1728])), T, int(rng.integers(1, 11)), dh))
    return out


ATTN_CASES = _attn_cases(10, 91)


@pytest.mark.parametrize('case', ATTN_CASES, ids=[f'B{c[0]}_L{c[1]}_T{c[2]}_h{c[3]}' for c in ATTN_CASES])
def test_bidirectional_attention_core_fuzz(case):
    """csrc/battn.hip forward + backward on ragged problems: voxel counts around the block's voxel capacity (256 / heads), a single voxel,
    every head count the kernel is built for (1..10: head x token tables beyond 256 entries at 10 heads x 27 tokens), both token / head-width
    builds; plus the squeeze-excite op on a ragged volume with the same seed."""
    B, L, T, heads, dh = case
    r = gc.check_battn(B, L, T, heads, dh, seed=100 + L)
    assert r['ok'], f"{r['name']}: err {r['err']:.3e} > tol {r['tol']:.1e}"
    C = 4 * heads * 2
    r = gc.check_squeeze_excite(C, (1 + L % 5, 2 + L % 3, 3), B)
    assert r['ok'], f"{r['name']}: err {r['err']:.3e} > tol {r['tol']:.1e}"
