"""Crop-directory loader against fixtures made by the reference's own `save()` / `load_augmented_data()` /
`estimate_tumor_volume()` (tests/golden/gen_golden_loader.py -> loader.npz).  CPU only: the loader is host code; the packed
hand-off to the device is covered by tests/test_gpu_edge.py::test_augmented_loader_feeds_packed_ingest."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from rsuper_amd.training import augmentation as aug
from rsuper_amd.training.dataset import augmented as A

CLASSES = synth.TINY_CLASSES
# float32 tolerance of the image after the intensity transforms (values are O(1..3)); the separable blur and the scalar
# min/max/mean reductions re-associate float32 sums, everything else is bit-identical
IMG_ATOL = 2e-5


def build_dir(tmp):
    img_list, lab_list, ufo = synth.loader_lists()
    for i, n in enumerate(synth.loader_names()):
        img, lab, unk, mask = synth.loader_crop(i, CLASSES)
        is_ufo = n in synth.LOADER_REPORTS
        A.save_crop(str(tmp), img_list[i], lab_list[i], img, lab, unk if is_ufo else None, mask if is_ufo else None,
                    {'tumor_in_crop': synth.LOADER_REPORTS[n]['tumor_in_crop']} if is_ufo else None,
                    synth.loader_report_rows(n) if is_ufo else None)
    return img_list, lab_list, ufo


def test_writer_layout_matches_reference_save(tmp_path, golden):
    g = golden['loader']
    build_dir(tmp_path)
    assert sorted(os.listdir(tmp_path)) == [str(f) for f in g['files']]    # generator asserted the bytes are identical too
    img_list, lab_list, _ = synth.loader_lists()
    p = A.crop_paths(str(tmp_path), img_list[2], lab_list[2])
    _, lab, unk, mask = synth.loader_crop(2, CLASSES)
    assert np.array_equal(np.load(p['label']), np.packbits(lab.astype(bool), axis=0))
    assert np.array_equal(np.load(p['unk']), np.packbits(unk.astype(bool), axis=0))
    assert np.array_equal(np.load(p['segment']), np.packbits(mask.astype(bool), axis=0))
    assert json.load(open(p['json']))['tumor_in_crop'] == 'pancreas'
    # the .npz list entry is stored as .npy
    assert os.path.exists(os.path.join(tmp_path, 'BDMAP_00000001.npy')) and os.path.exists(os.path.join(tmp_path, 'BDMAP_00000001_gt.npy'))


def test_estimate_tumor_volume_matches_reference(golden):
    g = golden['loader']
    for i, n in enumerate(synth.loader_names()):
        if n not in synth.LOADER_REPORTS:
            continue
        v, d = A.estimate_tumor_volume(synth.loader_report_rows(n), synth.LOADER_REPORTS[n]['tumor_in_crop'])
        assert len(v) == 10 and d.shape == (10, 3) and d.dtype == torch.float32
        assert np.array_equal(torch.tensor(v).float().numpy(), g['vol_%d' % i]), n
        assert np.array_equal(d.numpy(), g['diam_%d' % i]), n
    # known answers: sphere d=12, ellipsoid 5.5 x 6 x 7.25, 2-axis size takes the mean as third axis
    v, d = A.estimate_tumor_volume(synth.loader_report_rows('BDMAP_00000002'), 'pancreas')
    assert v[0] == pytest.approx(4 / 3 * np.pi * 6 ** 3) and d[1].tolist() == [10.0, 20.0, 15.0]
    assert sum(1 for x in v if x) == 3
    with pytest.raises(ValueError):
        A.estimate_tumor_volume([], 'spleen')
    with pytest.raises(ValueError):
        A.estimate_tumor_volume([], 7)


@pytest.mark.parametrize('name', ['brightness_multiply', 'brightness_additive', 'gamma', 'contrast', 'gaussian_blur',
                                  'gaussian_noise'])
def test_single_transform_matches_reference(golden, name):
    g = golden['loader']
    x = torch.from_numpy(synth.loader_crop(0, CLASSES)[0]).unsqueeze(0)
    call = {'brightness_multiply': lambda: aug.brightness_multiply(x, multiply_range=[0.7, 1.3]),
            'brightness_additive': lambda: aug.brightness_additive(x, std=0.1),
            'gamma': lambda: aug.gamma(x, gamma_range=[0.7, 1.5]),
            'contrast': lambda: aug.contrast(x, contrast_range=[0.7, 1.3]),
            'gaussian_blur': lambda: aug.gaussian_blur(x, sigma_range=[0.5, 1.5]),
            'gaussian_noise': lambda: aug.gaussian_noise(x, std=0.137)}[name]
    x0 = x.clone()
    for s in (11, 12):
        torch.manual_seed(s)
        y = call()
        ref = g['aug_%s_%d' % (name, s)]
        assert y.shape == x.shape and y.dtype == torch.float32
        assert np.abs(y.numpy() - ref).max() <= IMG_ATOL, name
    assert torch.equal(x, x0), 'input modified in place'


@pytest.mark.parametrize('packed', [False, True])
def test_load_matches_reference_load_augmented_data(tmp_path, golden, packed):
    g = golden['loader']
    img_list, lab_list, ufo = build_dir(tmp_path)
    ds = A.AugmentedCropDataset(str(tmp_path), img_list, lab_list, CLASSES, ufo_paths=ufo, packed=packed)
    assert len(ds) == 7
    seen = 0
    for k, s in enumerate(g['seeds']):
        i = int(g['load_%d_idx' % k])
        np.random.seed(int(s))
        torch.manual_seed(int(s))
        r = ds[i]
        # same number of draws consumed from both generators as the reference
        assert np.random.random() == float(g['load_%d_next_np' % k])
        assert torch.rand(1).numpy() == g['load_%d_next_torch' % k]
        _, lab, unk, mask = synth.loader_crop(i, CLASSES)
        is_ufo = synth.loader_names()[i] in synth.LOADER_REPORTS
        if not is_ufo:
            unk, mask = 0 * unk, 0 * mask
        assert r['image'].shape == (1,) + synth.LOADER_SHAPE and r['image'].dtype == torch.float32
        assert np.abs(r['image'].numpy() - g['load_%d_image' % k]).max() <= IMG_ATOL, (k, int(g['gates'][k]))
        assert np.array_equal(r['volumes'].numpy(), g['load_%d_volumes' % k])
        assert np.array_equal(r['diameters'].numpy(), g['load_%d_diameters' % k])
        if packed:
            for key, v in (('label', lab), ('unk_channels', unk), ('mask', mask)):
                assert r[key].dtype == torch.uint8 and r[key].shape == (1,) + synth.LOADER_SHAPE
                assert np.array_equal(np.unpackbits(r[key].numpy(), axis=0)[:len(CLASSES)], v), key
                assert not (r[key].numpy() & 0x07).any(), 'spare bits must be clear'
        else:
            assert r['label'].dtype == torch.uint8 and np.array_equal(r['label'].numpy(), lab)
            assert r['unk_channels'].dtype == torch.uint8 and np.array_equal(r['unk_channels'].numpy(), unk)
            assert r['mask'].dtype == torch.float32 and np.array_equal(r['mask'].numpy(), mask)
        seen |= int(g['gates'][k])
    assert seen == 0b111111, 'fixture must exercise all six transforms'


def test_default_collate_and_batch_keys(tmp_path):
    img_list, lab_list, ufo = build_dir(tmp_path)
    ds = A.AugmentedCropDataset(str(tmp_path), img_list, lab_list, CLASSES, ufo_paths=ufo, packed=True, augment=False)
    dl = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)
    b = next(iter(dl))
    assert set(b) == {'image', 'label', 'unk_channels', 'volumes', 'mask', 'diameters'}
    assert b['image'].shape == (3, 1) + synth.LOADER_SHAPE and b['label'].shape == (3, 1) + synth.LOADER_SHAPE
    assert b['volumes'].shape == (3, 10) and b['diameters'].shape == (3, 10, 3)
    assert torch.equal(b['image'][0], torch.from_numpy(synth.loader_crop(0, CLASSES)[0]))


def test_loader_errors(tmp_path):
    img_list, lab_list, ufo = build_dir(tmp_path)
    with pytest.raises(ValueError):
        A.AugmentedCropDataset(None, img_list, lab_list, CLASSES)
    ds = A.AugmentedCropDataset(str(tmp_path), img_list, lab_list, CLASSES, ufo_paths=ufo, augment=False)
    # report sample without its side files: raise, never rebuild silently
    p = A.crop_paths(str(tmp_path), img_list[3], lab_list[3])
    os.remove(p['unk'])
    with pytest.raises(FileNotFoundError):
        ds[3]
    # unknown voxels on an organ channel violate SanityAssertOutput
    _, lab, unk, mask = synth.loader_crop(4, CLASSES)
    unk[0, 0, 0, 0] = 1
    np.save(A.crop_paths(str(tmp_path), img_list[4], lab_list[4])['unk'], np.packbits(unk.astype(bool), axis=0))
    with pytest.raises(AssertionError):
        ds[4]
    dsp = A.AugmentedCropDataset(str(tmp_path), img_list, lab_list, CLASSES, ufo_paths=ufo, augment=False, packed=True)
    with pytest.raises(AssertionError):
        dsp[4]
    # wrong class count for the stored packing
    wide = A.AugmentedCropDataset(str(tmp_path), img_list, lab_list, synth.PANTS_CLASSES, ufo_paths=ufo, augment=False)
    with pytest.raises(AssertionError):
        wide[0]
    # unpacked (C, D, H, W) label files are accepted as the reference does (:1028)
    _, lab0, _, _ = synth.loader_crop(0, CLASSES)
    np.save(A.crop_paths(str(tmp_path), img_list[0], lab_list[0])['label'], lab0)
    assert np.array_equal(ds[0]['label'].numpy(), lab0)
    assert np.array_equal(np.unpackbits(dsp[0]['label'].numpy(), axis=0)[:len(CLASSES)], lab0)
