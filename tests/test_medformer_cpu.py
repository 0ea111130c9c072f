"""MedFormer: the CPU restatement oracle/medformer_oracle.py against fixtures of the reference class
(tests/golden/gen_golden_medformer.py -> medformer.npz)."""
import numpy as np
import torch

import synth
from oracle import medformer_oracle as mo

T = torch.from_numpy


def shapes_from_fixture(g):
    """Parameter shapes are not stored; the fixture lists names and sizes, the shapes come from the product's module tree."""
    from rsuper_amd.model.dim3.medformer import MedFormer
    cfg = synth.MEDFORMER_TINY
    net = MedFormer(1, len(synth.TINY_CLASSES), **{k: v for k, v in cfg.items() if k not in ('size', 'seed')})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert sorted(shapes) == [str(n) for n in g['param_names']], 'state_dict keys differ from the reference MedFormer'
    assert [int(np.prod(shapes[k])) for k in sorted(shapes)] == [int(n) for n in g['param_numel']]
    return shapes


def test_oracle_matches_reference_medformer(golden):
    g = golden['medformer']
    cfg = synth.MEDFORMER_TINY
    sdn = synth.fill_state_dict(shapes_from_fixture(g), cfg['seed'])
    assert abs(sum(float(np.abs(v).sum()) for v in sdn.values()) - g['param_checksum'][0]) < 1e-6 * g['param_checksum'][0]
    sd = {k: T(v).requires_grad_(True) for k, v in sdn.items()}
    y, aux = mo.medformer_forward(sd, T(synth.image(1, cfg['size'], seed=1234)), cfg)
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    ga = synth.rng(78).standard_normal(tuple(aux.shape)).astype(np.float32) / aux.numel()
    ((y * T(go)).sum() + (aux * T(ga)).sum()).backward()
    for nm, t in (('logits', y), ('aux', aux)):
        np.testing.assert_allclose(synth.subsample(t.detach().numpy(), 8192)[0], g[nm + '_sub'], atol=1e-4, rtol=1e-4, err_msg=nm)
        np.testing.assert_allclose(synth.summary(t.detach().numpy()), g[nm + '_summary'], rtol=1e-4)
    worst = 0.0
    for k in sd:
        ref = g[f'g_{k}_sub']
        scale = max(g[f'g_{k}_summary'][2], 1e-12)
        got = synth.subsample(sd[k].grad.numpy(), 1024)[0]
        worst = max(worst, float(np.abs(got - ref).max() / scale))
        np.testing.assert_allclose(got / scale, ref / scale, atol=1e-2, err_msg=k)
    assert worst < 1e-2
