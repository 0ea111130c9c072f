#!/usr/bin/env python3
"""Golden vectors for MedFormer from the UNMODIFIED reference class rsuper_train/model/dim3/medformer.py::MedFormer,
imported on CPU in the authoring container:

    python tests/golden/gen_golden_medformer.py   ->  tests/golden/medformer.npz

Tiny configuration with the structure of config/abdomenatlas_ufo/medformer_3d.yaml (conv stem + BasicBlocks at the two high
resolutions, bidirectional-attention stages below, semantic-map fusion, aux head; InstanceNorm + ReLU, depthwise projections).
Inputs and parameters are regenerated from tests/golden/synth.py seeds by the tests; only outputs are stored: strided samples and
summaries of both heads, of the encoder features and semantic maps, and of every parameter gradient.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = '/root/reference/rsuper_train'


def main():
    sys.path.insert(0, REF)
    for name, sub in (('model', 'model'), ('model.dim3', 'model/dim3')):       # bypass model/__init__ (MONAI / timm nets)
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    mf = importlib.import_module('model.dim3.medformer')
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    cfg = synth.MEDFORMER_TINY
    net = mf.MedFormer(1, len(synth.TINY_CLASSES), base_chan=cfg['base_chan'], map_size=cfg['map_size'], conv_block='BasicBlock',
                       conv_num=cfg['conv_num'], trans_num=cfg['trans_num'], chan_num=cfg['chan_num'], num_heads=cfg['num_heads'],
                       fusion_depth=cfg['fusion_depth'], fusion_dim=cfg['fusion_dim'], fusion_heads=cfg['fusion_heads'], expansion=4,
                       proj_type='depthwise', norm='in', act='relu', kernel_size=[[3, 3, 3]] * 5, scale=[[2, 2, 2]] * 4, aux_loss=True)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, cfg['seed'])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    out = {'param_checksum': np.array([sum(float(np.abs(v).sum()) for v in sd.values())], np.float64),
           'param_names': np.array(sorted(shapes)), 'param_numel': np.array([int(np.prod(shapes[k])) for k in sorted(shapes)])}

    feats = {}
    for name in ('inc', 'down1', 'down2', 'down3', 'down4', 'map_fusion', 'up1', 'up2', 'up3'):
        getattr(net, name).register_forward_hook(lambda mod, inp, o, name=name: feats.__setitem__(name, o))
    S = cfg['size']
    img = torch.from_numpy(synth.image(1, S, seed=1234))
    res = net(img)['segmentation']
    y, aux = res
    go = synth.rng(77).standard_normal(tuple(y.shape)).astype(np.float32) / y.numel()
    ga = synth.rng(78).standard_normal(tuple(aux.shape)).astype(np.float32) / aux.numel()
    (y * torch.from_numpy(go)).sum().add((aux * torch.from_numpy(ga)).sum()).backward()
    for nm, t in (('logits', y), ('aux', aux)):
        out[nm + '_sub'], _ = synth.subsample(t.detach().numpy(), 8192)
        out[nm + '_summary'] = synth.summary(t.detach().numpy())
    for name, o in feats.items():
        items = o if isinstance(o, (tuple, list)) else (o,)
        for j, t in enumerate(items):
            if t is not None:
                out[f'feat_{name}_{j}_summary'] = synth.summary(t.detach().numpy())
                out[f'feat_{name}_{j}_sub'], _ = synth.subsample(t.detach().numpy(), 1024)
    for k, v in net.named_parameters():
        g = v.grad.numpy()
        out[f'g_{k}_summary'] = synth.summary(g)
        out[f'g_{k}_sub'], _ = synth.subsample(g, 1024)
    path = os.path.join(HERE, 'medformer.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'kB;', len(shapes), 'tensors,', sum(int(np.prod(s)) for s in shapes.values()), 'parameters')


if __name__ == '__main__':
    main()
