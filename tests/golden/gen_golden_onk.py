#!/usr/bin/env python3
"""Golden vectors for the output-layer surgery of the public checkpoints, from the UNMODIFIED reference function
rsuper_train/model/dim3/medformer.py:224-319 `update_output_layer_onk` (call site train_ddp.py:574-580), imported on CPU in the authoring container:

    python tests/golden/gen_golden_onk.py   ->  tests/golden/onk.npz

The function only touches `model.outc`, `model.aux_out` (when `model.aux_loss`) and `model.classification_branch.head`, so the carrier is a
small nn.Module with those attributes (heads of the shipped geometry: 1x1x1 Conv3d with bias).  Old weights come from tests/golden/synth.py
seeds; rows of classes absent from the old list are the fresh nn.Conv3d initialisation under torch.manual_seed(SEED) (the test reproduces the
seed and the construction order: outc first, then aux_out)."""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = '/root/reference/rsuper_train'
OLD = sorted(['background', 'aorta', 'kidney_left', 'kidney_right', 'liver', 'pancreas', 'pancreatic_lesion', 'kidney_lesion', 'spleen'])
NEW = sorted(['background', 'liver', 'pancreas', 'pancreatic_lesion', 'pancreatic_pdac', 'pancreatic_pnet', 'pancreatic_cyst', 'spleen', 'colon_lesion', 'adrenal_gland_left'])
SEED = 1234
CASES = [('plain', False, True), ('copy_pancreas', True, True), ('no_aux', False, False), ('same', False, True)]


class Carrier(nn.Module):
    def __init__(self, c_out, c_aux, n, aux_loss):
        super().__init__()
        self.aux_loss = aux_loss
        self.outc = nn.Conv3d(c_out, n, kernel_size=1)
        self.aux_out = nn.Conv3d(c_aux, n, kernel_size=1)


def fill(model, seed):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed).items()})


def main():
    sys.path.insert(0, REF)
    for name, sub in (('model', 'model'), ('model.dim3', 'model/dim3')):       # bypass model/__init__ (MONAI / timm nets)
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    mf = importlib.import_module('model.dim3.medformer')
    out = {'old_classes': np.array(OLD), 'new_classes': np.array(NEW), 'seed': np.array([SEED])}
    for tag, copy_pancreas, aux in CASES:
        net = Carrier(32, 128, len(OLD), aux)
        fill(net, 11)
        new = OLD if tag == 'same' else NEW
        torch.manual_seed(SEED)
        ret = mf.update_output_layer_onk(net, original_classes=OLD, new_classes=new, copy_pancreas=copy_pancreas)
        assert ret is net
        for k, v in net.state_dict().items():
            out[f'{tag}.{k}'] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'onk.npz'), **out)
    print('wrote onk.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
