"""Deterministic synthetic inputs shared by the golden generator, the oracle tests,
the GPU parity tests and bench.py.  Pure numpy; never imports the reference.

Shapes/semantics follow SURVEY.md section 8(d):
  * image ~ N(0,1) clamped to [-3,3]  (reference CTs: clip[-991,500] -> z-score,
    dataset_conversion/nii2npz.py:64-74; train_ddp.py:312-313 asserts |x|<=100)
  * labels: multi-hot (B,C,D,H,W) organs as ellipsoids
  * "mask" sample : lesion annotated per voxel, unk = mask = 0, volumes = 0
  * "report" sample: lesion label 0, unk[lesion] = mask[lesion] = organ,
                     1-3 tumours with diameters/volumes
                     (training/dataset/dim3/dataset_abdomenatlas_UFO.py:1391-1407)
"""
import math
import numpy as np

TINY_CLASSES = ['kidney_left', 'kidney_lesion', 'kidney_right', 'pancreas', 'pancreatic_lesion']

# dataset_conversion/label_names_mask_dataset_pancreas.yaml (26 entries, alphabetical)
PANTS_CLASSES = [
    'adrenal_gland_left', 'adrenal_gland_right', 'aorta', 'bladder', 'colon', 'common_bile_duct',
    'duodenum', 'femur_left', 'femur_right', 'gall_bladder', 'kidney_left', 'kidney_right', 'liver',
    'lung_left', 'lung_right', 'pancreas', 'pancreas_body', 'pancreas_head', 'pancreas_tail',
    'pancreatic_lesion', 'postcava', 'prostate', 'spleen', 'stomach', 'superior_mesenteric_artery',
    'veins']


# dataset_conversion/label_names_mask_dataset.yaml (42 entries, alphabetical): BASELINE.json configs[4] (--classes_number 42)
MASK42_CLASSES = [
    'adrenal_gland_left', 'adrenal_gland_right', 'aorta', 'bladder', 'celiac_trunk', 'colon', 'common_bile_duct', 'duodenum',
    'esophagus', 'femur_left', 'femur_right', 'gall_bladder', 'hepatic_vessel', 'intestine', 'kidney_left', 'kidney_lesion',
    'kidney_right', 'liver', 'liver_lesion', 'liver_segment_1', 'liver_segment_2', 'liver_segment_3', 'liver_segment_4',
    'liver_segment_5', 'liver_segment_6', 'liver_segment_7', 'liver_segment_8', 'lung_left', 'lung_right', 'pancreas',
    'pancreas_body', 'pancreas_head', 'pancreas_tail', 'pancreatic_lesion', 'portal_vein_and_splenic_vein', 'postcava',
    'prostate', 'rectum', 'spleen', 'stomach', 'superior_mesenteric_artery', 'veins']


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def det_param(shape, seed, scale=None):
    """Deterministic parameter tensor: N(0, scale^2) with scale = 1/sqrt(fan_in)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else n
    if scale is None:
        scale = 1.0 / math.sqrt(max(fan_in, 1))
    return (rng(seed).standard_normal(n).astype(np.float32) * np.float32(scale)).reshape(shape)


def fill_state_dict(shapes, seed=0):
    """shapes: dict name -> shape.  Returns dict name -> float32 ndarray.
    Iterates in sorted-name order so reference module, oracle and product agree."""
    out = {}
    for i, name in enumerate(sorted(shapes)):
        out[name] = det_param(shapes[name], seed * 100003 + i)
    return out


def image(B, S, seed=1234):
    x = rng(seed).standard_normal((B, 1, S, S, S)).astype(np.float32)
    # add smooth structure so IN statistics are not trivially N(0,1)
    z = np.linspace(-1, 1, S, dtype=np.float32)
    x += 0.5 * np.sin(3.0 * z)[None, None, :, None, None] * np.cos(2.0 * z)[None, None, None, :, None]
    return np.clip(x, -3, 3).astype(np.float32)


def volume(shape, seed):
    """Non-cubic CT-like test volume (1, 1, D, H, W) for the sliding-window inference fixtures."""
    g = rng(seed)
    x = g.standard_normal((1, 1) + tuple(shape)).astype(np.float32)
    z = np.linspace(-1, 1, shape[0], dtype=np.float32)
    x += 0.5 * np.sin(3.0 * z)[None, None, :, None, None]
    return np.clip(x, -3, 3).astype(np.float32)


def _ellipsoid(S, center, radii):
    z, y, x = np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing='ij')
    d = ((z - center[0]) / radii[0]) ** 2 + ((y - center[1]) / radii[1]) ** 2 + ((x - center[2]) / radii[2]) ** 2
    return d <= 1.0


def batch(B, S, classes, kinds, seed=7, diam_range=(5.0, 12.0), max_tumors=2, n_tumors=None, diam_list=None, vol_list=None):
    """Build one synthetic mixed batch.

    kinds: list of 'mask' | 'report' | 'healthy' per sample.
    Returns dict of float32/uint8 ndarrays with the reference's batch keys
    (train_ddp.py:247-256): label, unk_channels, mask, volumes, diameters.
    n_tumors / diam_list / vol_list: fix the number of tumours of every report sample / their diameters / their reported volumes
    (default: the sphere of the diameter) instead of drawing them.
    """
    C = len(classes)
    g = rng(seed)
    label = np.zeros((B, C, S, S, S), np.uint8)
    unk = np.zeros_like(label)
    mask = np.zeros_like(label)
    volumes = np.zeros((B, 10), np.float32)
    diameters = np.zeros((B, 10, 3), np.float32)
    lesion_idx = [i for i, c in enumerate(classes) if 'lesion' in c]
    organ_of = {}
    for li in lesion_idx:
        organ = classes[li].split('_lesion')[0].replace('pancreatic', 'pancreas')
        cands = [i for i, c in enumerate(classes) if c == organ or c == organ + '_left']
        organ_of[li] = cands[0] if cands else None
    for b in range(B):
        # organs: random ellipsoids, one per non-lesion class
        for c in range(C):
            if c in lesion_idx:
                continue
            ctr = g.uniform(0.25 * S, 0.75 * S, 3)
            rad = g.uniform(S / 10.0, S / 5.0, 3)
            if classes[c] == 'pancreas':
                ctr = np.array([S / 2.0, S / 2.0, S / 2.0]) + g.uniform(-2, 2, 3)
                rad = np.array([S / 5.0] * 3)
            label[b, c] = _ellipsoid(S, ctr, rad)
        kind = kinds[b]
        li = lesion_idx[-1]  # pancreatic_lesion when present
        oi = organ_of[li]
        organ = label[b, oi].astype(bool)
        if kind == 'mask':
            ctr = np.array([S / 2.0] * 3) + g.uniform(-S / 12.0, S / 12.0, 3)
            r = g.uniform(3.0, max(3.5, S / 12.0))
            label[b, li] = _ellipsoid(S, ctr, (r, r, r)) & organ
        elif kind == 'report':
            unk[b, li] = organ
            mask[b, li] = organ
            nt = int(g.integers(1, max_tumors + 1)) if n_tumors is None else int(n_tumors)
            for t in range(nt):
                d = float(g.uniform(*diam_range)) if diam_list is None else float(diam_list[t])
                diameters[b, t] = (d, 0.8 * d, 0.7 * d)
                volumes[b, t] = (4.0 / 3.0) * math.pi * (d / 2.0) ** 3 if vol_list is None or vol_list[t] is None else float(vol_list[t])
        elif kind == 'healthy':
            pass
        else:
            raise ValueError(kind)
    return dict(label=label, unk_channels=unk, mask=mask, volumes=volumes, diameters=diameters)


# two sub-channels of one organ's lesion: get_lesion_channels max-merges them into 'pancreas_lesion'
# (training/losses_foundation.py:204-221)
MULTI_CH_CLASSES = ['kidney_left', 'kidney_right', 'pancreas', 'pancreatic_lesion_1', 'pancreatic_lesion_2']


def multi_ch_batch(B, S, kinds, seed=7, **kw):
    """batch() for MULTI_CH_CLASSES with the annotations spread over both lesion sub-channels: per-voxel tumours of 'mask'
    samples live in sub-channel 1, the unknown / report masks of 'report' samples are split between the two (lower half of
    the volume in sub-channel 1, upper half in sub-channel 2), so only the max-merge sees the whole organ."""
    bt = batch(B, S, MULTI_CH_CLASSES, kinds, seed=seed, **kw)
    c1, c2 = 3, 4
    for b, kind in enumerate(kinds):
        if kind == 'mask':
            bt['label'][b, c1] = bt['label'][b, c2]
            bt['label'][b, c2] = 0
        elif kind == 'report':
            for key in ('unk_channels', 'mask'):
                full = bt[key][b, c2].copy()
                bt[key][b, c1, :S // 2] = full[:S // 2]
                bt[key][b, c2, :S // 2] = 0
    return bt


def logits(B, C, S, seed=99, scale=2.0, smooth=True):
    """Continuous random logits (ties have measure zero -> selection ops are well defined)."""
    g = rng(seed)
    x = g.standard_normal((B, C, S, S, S)).astype(np.float32) * np.float32(scale)
    if smooth:
        # a bump near the centre so the ball search has a clear optimum
        z = (np.arange(S, dtype=np.float32) - S / 2.0) / (S / 6.0)
        bump = np.exp(-0.5 * (z[:, None, None] ** 2 + z[None, :, None] ** 2 + z[None, None, :] ** 2))
        x += 3.0 * bump[None, None]
    return x.astype(np.float32)


def subsample(a, n=4096):
    """Deterministic strided subsample of a flattened array (for compact fixtures)."""
    f = np.asarray(a).reshape(-1)
    step = max(1, f.size // n)
    return f[::step][:n].copy(), step


def summary(a):
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.array([f.sum(), (f * f).sum(), np.abs(f).max(), f.size], np.float64)


# ---- crop directories for the loader tests (dataset_abdomenatlas_UFO.py:937-1118) --------------------------------------
LOADER_SHAPE = (16, 20, 12)
LOADER_REPORTS = {
    # organ crop: rows are matched on 'Standardized Organ'
    'BDMAP_00000002': dict(tumor_in_crop='pancreas', rows=[
        dict(organ='pancreas', location='pancreas head', size='12'),
        dict(organ='pancreas', location='u', size='10 x 20'),
        dict(organ='kidney', location='kidney left', size='33'),
        dict(organ='pancreas', location='pancreas tail', size='5.5 x 6 x 7.25'),
        dict(organ='u', location='u', size='9'),
        dict(organ=float('nan'), location=float('nan'), size='4')]),
    # segment crop given as a list: rows are matched on 'Standardized Location', ' / ' lists must be fully inside
    'BDMAP_00000003': dict(tumor_in_crop=['head', 'body'], rows=[
        dict(organ='pancreas', location='head / body', size='8 x 4'),
        dict(organ='pancreas', location='head / tail', size='30'),
        dict(organ='pancreas', location='body', size='3.0'),
        dict(organ='pancreas', location='u', size='50'),
        dict(organ='pancreas', location=float('nan'), size='51')]),
    # crop not taken on a tumour
    'BDMAP_00000004': dict(tumor_in_crop='random', rows=[dict(organ='pancreas', location='pancreas head', size='12')]),
    'BDMAP_00000005': dict(tumor_in_crop=None, rows=[dict(organ='liver', location='liver segment 2', size='7 x 9')]),
    # organ crop given as a list; ' / ' lists in the organ column
    'BDMAP_00000006': dict(tumor_in_crop=['kidney left', 'kidney right'], rows=[
        dict(organ='kidney left', location='left', size='10'),
        dict(organ='kidney right / kidney left', location='right', size='2 x 3 x 4'),
        dict(organ='kidney', location='left', size='99')]),
}
LOADER_MASK_CASES = ['BDMAP_00000000', 'BDMAP_00000001']


def loader_names():
    return LOADER_MASK_CASES + sorted(LOADER_REPORTS)


def loader_lists(root='/data/npy'):
    """(img_list, lab_list, ufo_paths) with the reference's naming: <id>.npy / <id>_gt.npy; one entry is listed as .npz."""
    names = loader_names()
    img = ['%s/%s.npy' % (root, n) for n in names]
    lab = ['%s/%s_gt.npy' % (root, n) for n in names]
    img[1], lab[1] = img[1].replace('.npy', '.npz'), lab[1].replace('.npy', '.npz')
    return img, lab, [p for p, n in zip(img, names) if n in LOADER_REPORTS]


def loader_crop(i, classes):
    """Deterministic content of crop i: image (1,D,H,W) f32, label / unk / mask (C,D,H,W) uint8 0/1."""
    g = rng(4200 + i)
    D, H, W = LOADER_SHAPE
    C = len(classes)
    img = np.clip(g.standard_normal((1, D, H, W)).astype(np.float32) * 1.5, -3, 3)
    label = (g.random((C, D, H, W)) < 0.2).astype(np.uint8)
    lesion = [c for c, n in enumerate(classes) if 'lesion' in n]
    unk = np.zeros_like(label)
    mask = np.zeros_like(label)
    if loader_names()[i] in LOADER_REPORTS:
        for c in lesion:
            label[c] = 0
            unk[c] = (g.random((D, H, W)) < 0.3)
            mask[c] = (g.random((D, H, W)) < 0.25)
    return img, label, unk, mask


def loader_report_rows(name):
    return [{'BDMAP_ID': name, 'Standardized Organ': r['organ'], 'Standardized Location': r['location'],
             'Tumor Size (mm)': r['size']} for r in LOADER_REPORTS[name]['rows']]


# ---- MedFormer: tiny configuration shared by the golden generator, the oracle test and the GPU parity test
MEDFORMER_TINY = dict(base_chan=8, map_size=[2, 2, 2], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 1, 2, 1, 1, 1, 0, 0],
                      chan_num=[16, 32, 64, 80, 64, 32, 16, 8], num_heads=[1, 2, 4, 5, 4, 2, 1, 1], fusion_depth=1, fusion_dim=80,
                      fusion_heads=5, aux_loss=True, size=32, seed=5)


# ---- isolate_tumor / ball_loss at the diameters the benchmark's report batch draws (15-40 mm) ----------------------
# name -> (volume edge, diameter, reported volume, builder of x).  Volumes: the sphere of that diameter unless a case says otherwise.
def _sphere_vol(d):
    return (4.0 / 3.0) * math.pi * (d / 2.0) ** 3


def _sig(a):
    return (1.0 / (1.0 + np.exp(-a))).astype(np.float32)


def ball_case(name):
    """Input of one large-diameter isolate_tumor fixture: (x (S,S,S) float32 >= 0, diameter, tumour volume).
      d15 / d21 (48^3), d31 / d40 (64^3): smooth probability map with one optimum, ball kernels of edge 19 / 27 / 39 / 51
      border21: the optimum sits in a corner -> the inserted ball is clipped and the growth loop runs (losses_foundation.py:1450-1461)
      rewrite15: reported volume below the ball's own voxel count -> volume rewritten to nnz - 1 (:1431-1433)
      sparse21: only ~30 % of the voxels are positive, the rest exact zeros -> top-k runs out of positive voxels and the dilation round runs
                (:1513-1522); one round of the 7-ball saturates the masks to the whole inserted ball, so the result does not depend on WHICH
                zeros torch.topk picked (implementation-defined)
      organ31: support smaller than the reported tumour (sphere of radius 12 in 64^3, d = 31): the same situation without saturation -- the
               reference's result here depends on torch.topk's tie order among zeros (its CPU and GPU kernels differ); the fixture keeps it
               and the tests compare only the tie-independent part (the positive voxels)"""
    S = 64 if name in ('d31', 'd40', 'organ31') else 48
    lg = logits(1, 1, S, seed=5)[0, 0]
    x = _sig(lg)
    if name in ('d15', 'd21', 'd31', 'd40'):
        d = float(name[1:])
        return x, d, _sphere_vol(d)
    if name == 'border21':
        xb = (0.01 * x).astype(np.float32)
        xb[0:6, 0:6, 0:6] = 0.9
        return xb, 21.0, _sphere_vol(21.0)
    if name == 'rewrite15':
        return x, 15.0, 600.0
    if name == 'sparse21':
        keep = rng(77).random((S, S, S)) < 0.3
        return (x * keep).astype(np.float32), 21.0, _sphere_vol(21.0)
    if name == 'organ31':
        organ = _ellipsoid(S, (S / 2.0,) * 3, (12.0,) * 3)
        return (x * organ).astype(np.float32), 31.0, _sphere_vol(31.0)
    raise KeyError(name)


BALL_CASES = ['d15', 'd21', 'd31', 'd40', 'border21', 'rewrite15', 'sparse21', 'organ31']


# calculate_loss at 48^3 / 64^3 with three tumours per report sample (tests/golden/gen_golden_ball_large.py)
BALL_LOSS_CASES = {
    # tag: (edge, diameters of the three tumours of the report sample, their reported volumes (None: sphere), loss, deep supervision)
    'c48_both': (48, [31.0, 21.0, 15.0], None, 'ball_dice_both', False),
    'c48_deep_last': (48, [21.0, 15.0, 17.0], None, 'ball_dice_last', True),
    'c64_both': (64, [40.0, 21.0, 15.0], None, 'ball_dice_both', False),
    # a reported volume above what the ball of the reported diameter holds -> the growth loop (:1450-1461) runs inside calculate_loss,
    # i.e. the speculative device-side search must notice and repeat the sample with the exact search
    'c48_grow_both': (48, [15.0, 9.0, 21.0], [5000.0, None, None], 'ball_dice_both', False),
}


def ball_loss_case_inputs(tag):
    S, dl, vl, loss, deep = BALL_LOSS_CASES[tag]
    classes = TINY_CLASSES
    bt = batch(2, S, classes, ['mask', 'report'], seed=17, n_tumors=3, diam_list=dl, vol_list=vl)
    lg0 = logits(2, len(classes), S, seed=299)
    lg1 = logits(2, len(classes), S, seed=300)
    return classes, bt, lg0, lg1, loss, deep


# full-size cases (96^3, 26 classes): 'bench96' is EXACTLY the batch `bench.py --report` builds on rank 0 (two tumours, d = 18.6 / 24.2);
# 'full96_d40' has three tumours of d = 40 / 31 / 21 in the report sample
def fullsize_report_case(tag):
    classes = PANTS_CLASSES
    if tag == 'bench96':
        bt = batch(2, 96, classes, ['mask', 'report'], seed=7, diam_range=(5.0, 40.0), max_tumors=3)
    elif tag == 'full96_d40':
        bt = batch(2, 96, classes, ['mask', 'report'], seed=7, n_tumors=3, diam_list=[40.0, 31.0, 21.0])
    else:
        raise KeyError(tag)
    return classes, bt, logits(2, len(classes), 96, seed=12)


FULLSIZE_REPORT_CASES = ['bench96', 'full96_d40']


# two data-parallel ranks of the tiny training step, each on its own batch (tests/golden/gen_golden_ddp.py)
def ddp_rank_batch(rank, S=48):      # 48 like the tiny-UNet fixture: the bottom level then holds 3^3 voxels (at 32^3 it is 2^3 and InstanceNorm over 8 voxels
                                     # makes the fp32 gradients of the deepest layers move by 4 % with the summation order alone)
    return image(2, S, seed=4321 + rank), batch(2, S, TINY_CLASSES, ['mask', 'report'], seed=7 + rank, diam_range=(5.0, 9.0), max_tumors=2)
