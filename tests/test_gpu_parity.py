"""GPU parity tests proper (-m gpu): HIP path through the C ABI vs oracle / golden fixtures."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_checks as gc  # noqa: E402


def _loaded_native():
    from rsuper_amd.hip import lib
    lib.require_device()


@pytest.fixture(scope='module', autouse=True)
def native():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need an MI355X; the product path has no CPU fallback')
    _loaded_native()


@pytest.mark.parametrize('idx', range(len(gc.all_checks())))
def test_check(idx):
    fn, a = gc.all_checks()[idx]
    r = fn(*a)
    assert r['ok'], f"{r['name']}: err {r['err']:.3e} > tol {r['tol']:.1e} ({r['note']})"
