"""Pre-cropped ("augmented") crop directories: the storage format the training loop actually reads during a run, and the
loader for it.  Restates `AbdomenAtlasDataset.save` (rsuper_train/training/dataset/dim3/dataset_abdomenatlas_UFO.py:937-992),
`load_augmented_data` (:994-1118), `estimate_tumor_volume` (:1335-1415) and the shape/known-class checks of
`SanityAssertOutput` (:1417-1468).

Per crop `<name>` the directory holds
    <name>.npy                          float32 (1, D, H, W) CT crop
    <name>_gt.npy                       uint8 np.packbits(bool (C, D, H, W), axis=0) labels
    <name>_gt_unk.npy                   same packing, "lesion status unknown" voxels          (report samples)
    <name>_gt_chosen_tumor_segment.npy  same packing, organ / sub-segment the crop was taken on (report samples)
    <name>.json                         {"tumor_in_crop": <segment name | list | "random" | null>, ...}
    <name>.csv                          the report rows of the case ('Standardized Organ', 'Standardized Location',
                                        'Tumor Size (mm)')

`AugmentedCropDataset[idx]` returns the dictionary train_epoch consumes (train_ddp.py:247-262).  With `packed=True` the three
label-like volumes stay bit-packed (uint8 (ceil(C/8), D, H, W)) so that the batch crosses PCIe packed and is inflated on the
device by `ingest_packed_batch` (packed.py); with `packed=False` they are unpacked on the host exactly as the reference does.

Out of scope (offline preprocessing): building the crops from NIfTI volumes (`__getitem__`'s crop-on-tumor path,
`define_unknown_voxels`, `get_chosen_segment_mask`).  A report sample whose side files are missing therefore raises instead
of being regenerated.
"""
import json
import math
import os

import numpy as np
import torch
from torch.utils import data

from .. import augmentation

MAX_TUMORS = 10
_ORGAN_WORDS = ('liver', 'kidney', 'pancreas')
_SEGMENT_WORDS = ('segment', 'head', 'body', 'tail', 'left', 'right')


def _packed_channels(num_classes):
    return (num_classes + 7) // 8


def crop_paths(save_destination, img_path, lab_path):
    """File names of one crop from the original list entries (:1004-1012, :1081, :1094, :1097): basename, .npz -> .npy."""
    img = os.path.join(save_destination, os.path.basename(img_path).replace('.npz', '.npy'))
    lab = os.path.join(save_destination, os.path.basename(lab_path).replace('.npz', '.npy'))
    return {'image': img, 'label': lab,
            'unk': lab.replace('_gt.npy', '_gt_unk.npy'),
            'segment': lab.replace('.npy', '_chosen_tumor_segment.npy'),
            'json': img.replace('.npy', '.json'),
            'csv': img.replace('.npy', '.csv')}


def save_crop(save_destination, img_path, lab_path, image, label, unk_channels=None, mask=None, meta=None, report_rows=None):
    """Write one crop in the on-disk format above (the writer side, :937-992).  `image` (1, D, H, W) float; `label`,
    `unk_channels`, `mask` (C, D, H, W) 0/1; `meta` the json dict (needs 'tumor_in_crop' for report samples); `report_rows`
    a list of dicts or a DataFrame."""
    os.makedirs(save_destination, exist_ok=True)
    p = crop_paths(save_destination, img_path, lab_path)

    def packed(v):
        v = v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        return np.packbits(v.astype(np.bool_), axis=0)

    img = image.cpu().numpy() if isinstance(image, torch.Tensor) else np.asarray(image)
    np.save(p['image'], img)
    np.save(p['label'], packed(label))
    if unk_channels is not None:
        np.save(p['unk'], packed(unk_channels))
    if mask is not None:
        np.save(p['segment'], packed(mask))
    if meta is not None:
        with open(p['json'], 'w') as f:
            json.dump(meta, f)
    if report_rows is not None:
        import pandas as pd
        (report_rows if isinstance(report_rows, pd.DataFrame) else pd.DataFrame(list(report_rows))).to_csv(p['csv'], index=False)
    return p


def _rows(report):
    """Iterate report rows as mappings; accepts a DataFrame, a list of dicts or None."""
    if report is None:
        return None
    if hasattr(report, 'iterrows'):
        return [r for _, r in report.iterrows()]
    return list(report)


def estimate_tumor_volume(report, tumor_segment_crop):
    """Volumes (mm^3 == voxels at 1 mm spacing) and diameters of the reported tumours that lie in the organ / sub-segment the
    crop was taken on (:1335-1415).  Returns (list of 10 numbers, float32 tensor (10, 3)), zero padded.

    report: the case's rows with 'Standardized Organ', 'Standardized Location', 'Tumor Size (mm)'.
    tumor_segment_crop: None / 'random' (crop not on a tumour: all zeros), a name, or a list of names.
    A row counts when every ' / '-separated part of its location (organ column for organ crops, location column for
    segment crops) is in tumor_segment_crop; sizes are 'd' (sphere), 'a x b' (third axis = mean) or 'a x b x c' (ellipsoid)."""
    if tumor_segment_crop is None or tumor_segment_crop == 'random':
        return [0] * MAX_TUMORS, torch.zeros((MAX_TUMORS, 3)).float()
    if isinstance(tumor_segment_crop, str):
        tumor_segment_crop = [tumor_segment_crop]
    elif not isinstance(tumor_segment_crop, list):
        raise ValueError('tumor_segment_crop must be a list or a string.')
    joined = ''.join(tumor_segment_crop)
    if any(w in joined for w in _ORGAN_WORDS):
        col = 'Standardized Organ'
    elif any(w in joined for w in _SEGMENT_WORDS):
        col = 'Standardized Location'
    else:
        raise ValueError('tumor_segment_crop does not contain organs or segments:', tumor_segment_crop)
    rows = _rows(report)
    if rows is None:
        raise ValueError('crop was taken on %s but the case has no report rows' % (tumor_segment_crop,))

    sizes_in_crop = []
    for row in rows:
        location = row[col]
        if not isinstance(location, str) or location.lower() == 'u':
            continue
        parts = location.split(' / ') if '/' in location else [location]
        if all(part in tumor_segment_crop for part in parts):
            sizes_in_crop.append(row['Tumor Size (mm)'])

    volumes, diameters = [], []
    for size in sizes_in_crop:
        if 'x' not in size:
            axes = [float(size)] * 3
            volumes.append((4 / 3) * math.pi * ((axes[0] / 2) ** 3))
        else:
            axes = [float(s) for s in size.split(' x ')]
            if len(axes) == 2:
                axes.append(sum(axes) / 2)
            volumes.append((4 / 3) * math.pi * ((axes[0] / 2) * (axes[1] / 2) * (axes[2] / 2)))
        diameters.append(axes)
    for _ in range(len(volumes), MAX_TUMORS):
        volumes.append(0)
        diameters.append([0, 0, 0])
    return volumes, torch.tensor(diameters).float()


def online_intensity_augmentation(img):
    """The six gated transforms of :1047-1060 in the reference's order.  Gates come from numpy's global generator
    (`np.random.random() < 0.3`), parameters from torch's, the noise std from numpy's again."""
    if np.random.random() < 0.3:
        img = augmentation.brightness_multiply(img, multiply_range=[0.7, 1.3])
    if np.random.random() < 0.3:
        img = augmentation.brightness_additive(img, std=0.1)
    if np.random.random() < 0.3:
        img = augmentation.gamma(img, gamma_range=[0.7, 1.5])
    if np.random.random() < 0.3:
        img = augmentation.contrast(img, contrast_range=[0.7, 1.3])
    if np.random.random() < 0.3:
        img = augmentation.gaussian_blur(img, sigma_range=[0.5, 1.5])
    if np.random.random() < 0.3:
        std = np.random.random() * 0.2
        img = augmentation.gaussian_noise(img, std=std)
    return img


def check_sample(classes, classes_ufo, label, unk_channels, mask):
    """Shape and known-class invariants of `SanityAssertOutput` (:1421-1466) on unpacked (C, D, H, W) volumes: unknown voxels
    and the segment mask may be set only on lesion channels and on classes the report dataset lacks."""
    classes = sorted(classes)
    assert label.dim() == 4, 'tensor_lab must have 4 dimensions'
    assert label.shape[0] == len(classes), 'label has %d channels, expected %d' % (label.shape[0], len(classes))
    assert unk_channels.shape == label.shape and mask.shape == label.shape, \
        'label, unk_channels and mask must have the same shape: %s %s %s' % (label.shape, unk_channels.shape, mask.shape)
    missing = set(classes) - set(classes_ufo) - {'liver', 'pancreas'}
    known = [i for i, c in enumerate(classes) if not ('lesion' in c.lower() or c in missing)]
    assert unk_channels[known].sum().item() == 0, 'unknown voxels on a non-lesion class'
    assert mask[known].sum().item() == 0, 'segment mask on a non-lesion class'


class AugmentedCropDataset(data.Dataset):
    """Training-mode reader of a crop directory (`load_augmented=True` in the reference, :994-1118).

    img_list / lab_list   the original per-case paths (only their basenames name the crop files)
    classes               channel names in label order (must be sorted, as SanityAssertOutput assumes)
    ufo_paths             the entries of img_list that are report-supervised; all others are per-voxel annotated
    reports               optional callable idx -> report rows (DataFrame / list of dicts / None); default: `<name>.csv`
    classes_ufo           classes the report dataset has organ masks for (default: all)
    packed                keep label / unk_channels / mask bit-packed for `ingest_packed_batch`
    augment               apply the online intensity augmentations (reference: always in train mode)
    """

    def __init__(self, save_destination, img_list, lab_list, classes, ufo_paths=(), reports=None, classes_ufo=None,
                 packed=False, augment=True, check=True):
        if save_destination is None:
            raise ValueError('load_augmented=True but save_destination=None. Cannot load augmented data.')
        assert len(img_list) == len(lab_list)
        self.save_destination = save_destination
        self.img_list, self.lab_list = list(img_list), list(lab_list)
        self.classes = list(classes)
        self.num_classes = len(self.classes)
        self.ufo_paths = set(ufo_paths)
        self.reports = reports
        self.classes_ufo = list(classes_ufo) if classes_ufo is not None else list(self.classes)
        self.packed, self.augment, self.check = bool(packed), bool(augment), bool(check)

    @classmethod
    def from_directory(cls, save_destination, classes, **kw):
        """Dataset over every crop found in `save_destination`: `<name>.npy` + `<name>_gt.npy` pairs in sorted order.  A crop with a
        `<name>.json` beside it is LOADED through the report path (the reference derives that split from its data lists, :244-286,
        offline artefacts that are not read here).  The reference's save() writes the .json for every crop, so in a directory it
        produced all crops take that path -- equivalent for per-voxel crops: their `tumor_in_crop` is null and their side masks are
        zero, which is what the mask path yields."""
        files = set(os.listdir(save_destination))
        names = sorted(f[:-len('_gt.npy')] for f in files if f.endswith('_gt.npy') and f[:-len('_gt.npy')] + '.npy' in files)
        img_list = [os.path.join(save_destination, n + '.npy') for n in names]
        lab_list = [os.path.join(save_destination, n + '_gt.npy') for n in names]
        ufo = [p for p, n in zip(img_list, names) if n + '.json' in files]
        return cls(save_destination, img_list, lab_list, classes, ufo_paths=ufo, **kw)

    def __len__(self):
        return len(self.img_list)

    # -- storage -----------------------------------------------------------------------------------------------------
    def _load_bits(self, path):
        """uint8 array as stored: either already (C, D, H, W) 0/1 or bit-packed along axis 0 (:1028-1036)."""
        a = np.load(path, allow_pickle=False)
        if a.shape[0] == self.num_classes:
            return a, False
        assert a.shape[0] * 8 < self.num_classes + 10 and a.shape[0] * 8 >= self.num_classes, \
            'packed channel count %d does not fit %d classes' % (a.shape[0], self.num_classes)
        return a, True

    def _unpacked(self, a, is_packed):
        return np.unpackbits(a, axis=0)[:self.num_classes] if is_packed else a

    def _packed(self, a, is_packed):
        if is_packed:
            # bits past num_classes are dropped by the reference's [:num_classes]; clear them so both forms agree
            spare = a.shape[0] * 8 - self.num_classes
            if spare:
                a = a.copy()
                a[-1] &= np.uint8((0xFF << spare) & 0xFF)
            return a
        return np.packbits(a.astype(np.bool_), axis=0)

    def _report(self, idx, paths):
        if self.reports is not None:
            return self.reports(idx)
        if not os.path.exists(paths['csv']):
            return None
        import pandas as pd
        return pd.read_csv(paths['csv'], dtype=str)    # sizes such as '12' must stay text ('x' in size, float(size))

    def _check(self, lab, lab_packed, unk, seg):
        """`check_sample`, evaluated on the packed bytes when that is how the volumes are stored (channel 8p+j is bit 7-j
        of packed plane p), so the packed path never inflates on the host."""
        if lab_packed and unk[1] and seg[1]:
            assert unk[0].shape == lab.shape and seg[0].shape == lab.shape, 'label, unk_channels and mask shapes differ'
            classes = sorted(self.classes)
            missing = set(classes) - set(self.classes_ufo) - {'liver', 'pancreas'}
            kmask = np.zeros((lab.shape[0],), np.uint8)
            for i, c in enumerate(classes):
                if not ('lesion' in c.lower() or c in missing):
                    kmask[i // 8] |= np.uint8(0x80 >> (i % 8))
            kmask = kmask.reshape((-1,) + (1,) * (lab.ndim - 1))
            assert not (unk[0] & kmask).any(), 'unknown voxels on a non-lesion class'
            assert not (seg[0] & kmask).any(), 'segment mask on a non-lesion class'
        else:
            t = lambda e: torch.from_numpy(self._unpacked(*e))
            check_sample(self.classes, self.classes_ufo, t((lab, lab_packed)), t(unk), t(seg))

    # -- one sample --------------------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        p = crop_paths(self.save_destination, self.img_list[idx], self.lab_list[idx])
        img = torch.from_numpy(np.load(p['image'], allow_pickle=False)).float()
        if img.dim() != 4 or img.shape[0] != 1:
            raise ValueError('%s: expected a (1, D, H, W) crop, got %s' % (p['image'], tuple(img.shape)))
        lab, lab_packed = self._load_bits(p['label'])

        if self.augment:
            img = online_intensity_augmentation(img.unsqueeze(0)).squeeze(0)

        vol_shape = (self.num_classes,) + tuple(lab.shape[1:])
        if self.img_list[idx] not in self.ufo_paths:
            # annotated per voxel: nothing unknown, no report supervision (:1073-1078)
            unk = seg = None
            volumes = [0] * MAX_TUMORS
            diameters = torch.zeros((MAX_TUMORS, 3)).float()
        else:
            for key in ('unk', 'segment', 'json'):
                if not os.path.exists(p[key]):
                    raise FileNotFoundError('%s is missing: report samples need the side files written with the crop '
                                            '(rebuilding them from the full CT is offline preprocessing)' % p[key])
            unk = self._load_bits(p['unk'])
            seg = self._load_bits(p['segment'])
            with open(p['json']) as f:
                meta = json.load(f)
            volumes, diameters = estimate_tumor_volume(self._report(idx, p), meta['tumor_in_crop'])

        def volume(entry, want_packed):
            if entry is None:
                shape = ((_packed_channels(self.num_classes),) + vol_shape[1:]) if want_packed else vol_shape
                return torch.zeros(shape, dtype=torch.uint8)
            a, is_packed = entry
            return torch.from_numpy(np.ascontiguousarray(self._packed(a, is_packed) if want_packed
                                                         else self._unpacked(a, is_packed)))

        if self.check and unk is not None:
            self._check(lab, lab_packed, unk, seg)

        out = {'image': img,
               'label': volume((lab, lab_packed), self.packed),
               'unk_channels': volume(unk, self.packed),
               'volumes': torch.tensor(volumes).float(),
               'diameters': diameters.type_as(img)}
        out['mask'] = volume(seg, self.packed) if self.packed else volume(seg, False).float()
        return out
