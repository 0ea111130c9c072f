#!/usr/bin/env python3
"""Golden vectors for the crop-directory loader, produced by the UNMODIFIED reference class
`AbdomenAtlasDataset` (/root/reference/rsuper_train/training/dataset/dim3/dataset_abdomenatlas_UFO.py): its `save()` writes the
crop directory, its `load_augmented_data()` reads it back under seeded numpy / torch generators, its `estimate_tumor_volume()`
prices the report rows.  Run in the authoring container only:

    python tests/golden/gen_golden_loader.py

Import shims: `SimpleITK`, `torchvision`, `nibabel` are absent here and are only touched by NIfTI debug dumps
and 2-D transforms outside this path, so empty modules are registered for them.  The dataset object is created without
running __init__ (which walks the real data lists); only the attributes `load_augmented_data` reads are set, and
`get_tumor_segment_labels` returns the rows of tests/golden/synth.LOADER_REPORTS.

Writes tests/golden/loader.npz: expected outputs only; the inputs are regenerated from synth.py seeds by the tests.
"""
import os
import sys
import types
import tempfile
import importlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth  # noqa: E402

REF = '/root/reference/rsuper_train'
CLASSES = synth.TINY_CLASSES


def import_reference():
    for name in ('SimpleITK', 'nibabel', 'torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.path.insert(0, REF)
    ds = importlib.import_module('training.dataset.dim3.dataset_abdomenatlas_UFO')
    aug = importlib.import_module('training.augmentation')
    return ds, aug


def gates(seed):
    """Which of the six transforms fire for this numpy seed (the loader's own draw order)."""
    st = np.random.RandomState(seed)
    fired = []
    for i in range(6):
        if st.random_sample() < 0.3:
            fired.append(i)
            if i == 5:
                st.random_sample()
    return fired


def pick_seeds():
    """Smallest seeds that (together) fire every transform, plus one that fires none and one that fires three or more."""
    chosen, covered = [], set()
    for s in range(1000):
        f = gates(s)
        if set(f) - covered:
            chosen.append(s)
            covered |= set(f)
        if len(covered) == 6:
            break
    chosen.append(next(s for s in range(1000) if not gates(s)))
    chosen.append(next(s for s in range(1000) if len(gates(s)) >= 3 and s not in chosen))
    return chosen


def main():
    import pandas as pd
    refds, refaug = import_reference()
    out = {}
    names = synth.loader_names()
    img_list, lab_list, ufo = synth.loader_lists()

    ds = object.__new__(refds.AbdomenAtlasDataset)
    ds.img_list, ds.lab_list, ds.UFO_paths = img_list, lab_list, ufo
    ds.classes, ds.num_classes, ds.classes_UFO = CLASSES, len(CLASSES), CLASSES
    ds.mode, ds.generate_pair = 'train', None
    ds.counter, ds.save_counter = 99, 0
    ds.current_sample = 'BDMAP_00000000.npy'
    frames = {n: pd.DataFrame(synth.loader_report_rows(n)) for n in synth.LOADER_REPORTS}
    ds.get_tumor_segment_labels = lambda idx: (None, frames.get(names[idx]))

    with tempfile.TemporaryDirectory() as tmp, tempfile.TemporaryDirectory() as tmp2:
        ds.save_destination = tmp
        # writer: the reference's save() against this repo's save_crop()
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE))))
        rs = importlib.import_module('rsuper_amd.training.dataset.augmented')
        for i, n in enumerate(names):
            img, lab, unk, mask = synth.loader_crop(i, CLASSES)
            is_ufo = n in synth.LOADER_REPORTS
            meta = {'tumor_in_crop': synth.LOADER_REPORTS[n]['tumor_in_crop']} if is_ufo else None
            ds.save(torch.from_numpy(img), torch.from_numpy(lab), i, tumor_dict=frames.get(n), dta=meta,
                    unk_channels_tensor=torch.from_numpy(unk) if is_ufo else None,
                    chosen_segment_mask=torch.from_numpy(mask) if is_ufo else None)
            rs.save_crop(tmp2, img_list[i], lab_list[i], img, lab, unk if is_ufo else None, mask if is_ufo else None,
                         meta, frames.get(n))
        mine, theirs = sorted(os.listdir(tmp2)), sorted(os.listdir(tmp))
        assert mine == theirs, (mine, theirs)
        for f in mine:
            a, b = open(os.path.join(tmp, f), 'rb').read(), open(os.path.join(tmp2, f), 'rb').read()
            assert a == b, 'writer mismatch in ' + f
        out['files'] = np.array(theirs)
        print('writer: %d files byte-identical to the reference save()' % len(mine))

        # estimate_tumor_volume on every report case
        for i, n in enumerate(names):
            if n in synth.LOADER_REPORTS:
                v, d = ds.estimate_tumor_volume(i, synth.LOADER_REPORTS[n]['tumor_in_crop'])
                out['vol_%d' % i] = torch.tensor(v).float().numpy()
                out['diam_%d' % i] = d.numpy()

        # load_augmented_data under seeded generators
        seeds = pick_seeds()
        seeds += [100 + k for k in range(len(seeds), len(names))]   # every crop is loaded at least once
        out['seeds'] = np.array(seeds)
        out['gates'] = np.array([sum(1 << g for g in gates(s)) for s in seeds])
        for k, s in enumerate(seeds):
            i = k % len(names)
            np.random.seed(s)
            torch.manual_seed(s)
            r = ds.load_augmented_data(i)
            img, lab, unk, mask = synth.loader_crop(i, CLASSES)
            is_ufo = names[i] in synth.LOADER_REPORTS
            assert r['label'].dtype == torch.uint8 and np.array_equal(r['label'].numpy(), lab)
            assert np.array_equal(r['unk_channels'].numpy(), unk if is_ufo else 0 * unk)
            assert r['mask'].dtype == torch.float32 and np.array_equal(r['mask'].numpy(), (mask if is_ufo else 0 * mask))
            out['load_%d_idx' % k] = np.array(i)
            out['load_%d_image' % k] = r['image'].numpy()
            out['load_%d_volumes' % k] = r['volumes'].numpy()
            out['load_%d_diameters' % k] = r['diameters'].numpy()
            # the stream position afterwards pins the number of draws consumed
            out['load_%d_next_np' % k] = np.array(np.random.random())
            out['load_%d_next_torch' % k] = torch.rand(1).numpy()
            print('seed %d idx %d gates %s' % (s, i, gates(s)))

    # each transform on its own, fixed torch seed
    x = torch.from_numpy(synth.loader_crop(0, CLASSES)[0]).unsqueeze(0)
    single = {
        'brightness_multiply': lambda: refaug.brightness_multiply(x, multiply_range=[0.7, 1.3]),
        'brightness_additive': lambda: refaug.brightness_additive(x, std=0.1),
        'gamma': lambda: refaug.gamma(x.clone(), gamma_range=[0.7, 1.5]),
        'contrast': lambda: refaug.contrast(x, contrast_range=[0.7, 1.3]),
        'gaussian_blur': lambda: refaug.gaussian_blur(x, sigma_range=[0.5, 1.5]),
        'gaussian_noise': lambda: refaug.gaussian_noise(x, std=0.137),
    }
    for name, fn in single.items():
        for s in (11, 12):
            torch.manual_seed(s)
            out['aug_%s_%d' % (name, s)] = fn().numpy()

    path = os.path.join(HERE, 'loader.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'kB')


if __name__ == '__main__':
    main()
