"""Synthetic stand-in for the UFO / AbdomenAtlas training set: yields the batch dictionary train_epoch reads
(rsuper_train/train_ddp.py:247-262 -- image, label, unk_channels, volumes, mask, diameters[, name]) with the shapes and dtypes
of training/dataset/dim3/dataset_abdomenatlas_UFO.py:__getitem__.  Used by the smoke run of `python -m rsuper_amd.train_ddp
--synthetic` and by the driver tests; real crops come from `load_augmented_data` (dataset_abdomenatlas_UFO.py:994-1118)."""
import math

import numpy as np
import torch
from torch.utils import data


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _ellipsoid(S, center, radii):
    z, y, x = np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing='ij')
    return (((z - center[0]) / radii[0]) ** 2 + ((y - center[1]) / radii[1]) ** 2 + ((x - center[2]) / radii[2]) ** 2) <= 1.0


class SyntheticUFODataset(data.Dataset):
    """`length` deterministic samples of size S^3 over `classes`; even indices are per-voxel annotated ('mask') samples, odd
    indices report-only samples (lesion label 0, unknown / segment mask = the organ, 1-3 tumours with diameters and volumes),
    the 50/50 source balance of the reference loader (:192-202)."""

    def __init__(self, classes, size=96, length=64, seed=0, packed=False):
        self.classes, self.S, self.length, self.seed = list(classes), int(size), int(length), int(seed)
        # packed: label / unk_channels / mask leave as np.packbits(axis = class), the form AugmentedCropDataset(packed=True) yields (train_epoch then
        # inflates them on the device or hands the packed label to the loss kernels, dataset/packed.py)
        self.packed = bool(packed)
        self.img_list = list(range(self.length))
        self.lesion = [i for i, c in enumerate(self.classes) if 'lesion' in c]
        self.organ = {}
        for li in self.lesion:
            name = self.classes[li].split('_lesion')[0].replace('pancreatic', 'pancreas')
            cands = [i for i, c in enumerate(self.classes) if c == name or c == name + '_left']
            self.organ[li] = cands[0] if cands else None

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        S, C = self.S, len(self.classes)
        g = _rng(self.seed * 100003 + int(idx))
        img = np.clip(g.standard_normal((1, S, S, S)).astype(np.float32), -3, 3)
        label = np.zeros((C, S, S, S), np.uint8)
        unk = np.zeros_like(label)
        mask = np.zeros_like(label)
        volumes = np.zeros((10,), np.float32)
        diameters = np.zeros((10, 3), np.float32)
        for c in range(C):
            if c in self.lesion:
                continue
            label[c] = _ellipsoid(S, g.uniform(0.25 * S, 0.75 * S, 3), g.uniform(S / 10.0, S / 5.0, 3))
        usable = [li for li in self.lesion if self.organ[li] is not None]
        if usable:
            li = usable[-1]
            organ = label[self.organ[li]].astype(bool)
            if idx % 2 == 0:
                r = g.uniform(3.0, max(3.5, S / 12.0))
                label[li] = _ellipsoid(S, np.array([S / 2.0] * 3) + g.uniform(-S / 12.0, S / 12.0, 3), (r, r, r)) & organ
            else:
                unk[li] = organ
                mask[li] = organ
                for t in range(int(g.integers(1, 4))):
                    d = float(g.uniform(5.0, 40.0))
                    diameters[t] = (d, 0.8 * d, 0.7 * d)
                    volumes[t] = (4.0 / 3.0) * math.pi * (d / 2.0) ** 3
        if self.packed:
            label, unk, mask = (np.packbits(v.astype(np.bool_), axis=0) for v in (label, unk, mask))
        return {'image': torch.from_numpy(img), 'label': torch.from_numpy(label), 'unk_channels': torch.from_numpy(unk),
                'volumes': torch.from_numpy(volumes), 'mask': torch.from_numpy(mask), 'diameters': torch.from_numpy(diameters),
                'name': f'synthetic_{idx:05d}'}
