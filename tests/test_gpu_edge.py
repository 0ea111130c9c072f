"""GPU edge cases: odd / ragged spatial sizes, batch 1, the 128^3 / 42-class shape of BASELINE config 5, empty report
information, and the reference's error behaviour (ValueError / AssertionError / NotImplementedError, never a fallback)."""
import argparse
import math

import numpy as np
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import synth  # noqa: E402
import gpu_checks as gc  # noqa: E402

DEV = 'cuda'


@pytest.fixture(scope='module', autouse=True)
def native():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need an MI355X; the product path has no CPU fallback')
    from rsuper_amd.hip import lib
    lib.require_device()


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(7, 9, 11), (5, 4, 17), (2, 2, 2), (3, 16, 33)])
def test_conv_ragged_sizes(mode, shape):
    """Spatial sizes that are not multiples of the 4x4x16 tile (partial tiles on every axis, single voxel)."""
    r = gc.check_conv_fwd(mode, 1, shape, 8, 8, 16, True, False, seed=hash(shape) % 97)
    assert r['ok'], r
    r = gc.check_conv_bwd(mode, 1, shape, 16, 0, 8, True, seed=hash(shape) % 89)
    assert r['ok'], r


def test_maxpool_odd_sizes_match_torch():
    from rsuper_amd.hip import ops
    x = torch.randn((1, 8, 7, 9, 5))
    xr = x.clone().requires_grad_(True)
    y = F.max_pool3d(xr, 2)
    go = torch.randn_like(y)
    y.backward(go)
    xc = gc.to_cl(x, torch.float32).requires_grad_(True)
    yo, _ = ops.MaxPoolFn.apply(xc)
    yo.backward(gc.to_cl(go, torch.float32))
    assert torch.allclose(gc.from_cl(yo.detach()), y.detach(), atol=1e-6)
    assert torch.allclose(gc.from_cl(xc.grad), xr.grad, atol=1e-6)


def test_unet_batch1_and_config5_shape():
    """B=1, 128^3, 42 classes (BASELINE config 5 shape) runs forward+backward in bf16 with finite values."""
    from rsuper_amd.model.dim3.unet import UNet
    torch.manual_seed(0)
    net = UNet(1, 32, num_classes=42, compute_dtype='bf16').to(DEV)
    img = torch.from_numpy(synth.image(1, 128, seed=3)).to(DEV)
    y = net(img)['segmentation']
    assert y.shape == (1, 42, 128, 128, 128) and bool(torch.isfinite(y).all())
    y.square().mean().backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())


def _args(**kw):
    d = dict(loss='ball_dice_both', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
             ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
             classification_branch=False)
    d.update(kw)
    return argparse.Namespace(**d)


def test_calculate_loss_error_behaviour():
    from rsuper_amd.training import losses_foundation as lf
    classes = synth.TINY_CLASSES
    S = 16
    bt = synth.batch(2, S, classes, ['mask', 'report'], seed=7, diam_range=(4.0, 6.0))
    t = {k: torch.from_numpy(v).to(DEV) for k, v in bt.items()}
    lg = torch.from_numpy(synth.logits(2, len(classes), S, seed=1)).to(DEV).requires_grad_(True)
    # inconsistent report sample: segment mask set but unk all zero (losses_foundation.py:866-867)
    with pytest.raises(ValueError, match='unk_voxels should not be all zeros'):
        lf.calculate_loss({'segmentation': lg}, t['label'], torch.zeros_like(t['unk_channels']), _args(), None, t['mask'], t['volumes'],
                          t['diameters'], classes)
    # ... or no report volume (:868-869)
    with pytest.raises(ValueError, match='tumor_volumes_report should not be all zeros'):
        lf.calculate_loss({'segmentation': lg}, t['label'], t['unk_channels'], _args(), None, t['mask'], torch.zeros_like(t['volumes']),
                          t['diameters'], classes)
    # wrong class list length (:872)
    with pytest.raises(AssertionError):
        lf.calculate_loss({'segmentation': lg}, t['label'], t['unk_channels'], _args(), None, t['mask'], t['volumes'], t['diameters'], classes[:-1])
    # NaN guard (:1070-1071)
    bad = lg.detach().clone()
    bad[0, 0, 0, 0, 0] = float('nan')
    with pytest.raises(ValueError, match='loss is nan'):
        lf.calculate_loss({'segmentation': bad.requires_grad_(True)}, t['label'], t['unk_channels'], _args(report_volume_loss_basic=0.0), None,
                          t['mask'], t['volumes'], t['diameters'], classes)
    # baselines outside the accelerated path are rejected loudly, not emulated
    with pytest.raises(NotImplementedError):
        lf.calculate_loss({'segmentation': lg}, t['label'], t['unk_channels'], _args(), None, t['mask'], t['volumes'], t['diameters'], classes,
                          model_genesis=True)


def test_calculate_loss_without_lesion_classes_and_without_unk():
    """No lesion channel in the class list -> only the segmentation term ('report' key, as the reference's scalar path);
    unk_voxels=None -> every voxel known."""
    from rsuper_amd.training import losses_foundation as lf
    from oracle import losses_oracle as lo
    classes = ['kidney_left', 'kidney_right', 'pancreas']
    S = 16
    g = synth.rng(5)
    label = torch.from_numpy((g.random((2, 3, S, S, S)) < 0.2).astype(np.uint8))
    lg = torch.from_numpy(synth.logits(2, 3, S, seed=2))
    x = lg.to(DEV).requires_grad_(True)
    res = lf.calculate_loss({'segmentation': x}, label.to(DEV), None, _args(report_volume_loss_basic=0.0), None, None,
                            torch.zeros(2, 10, device=DEV), torch.zeros(2, 10, 3, device=DEV), classes)
    res['overall'].backward()
    ref = lo.calculate_loss({'segmentation': lg.clone().requires_grad_(True)}, label, torch.zeros_like(label), _args(report_volume_loss_basic=0.0),
                            torch.zeros_like(label), torch.zeros(2, 10), torch.zeros(2, 10, 3), classes)
    assert set(res) == {'segmentation', 'report', 'overall'}
    assert abs(float(res['overall'].detach()) - float(ref['overall'].detach())) < 1e-4


def test_checkpoint_roundtrip(tmp_path):
    """Checkpoint dict keys of train_ddp.py:184-189; plain state_dicts that load back (and into a fresh module)."""
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import save_checkpoint, load_checkpoint, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    net = UNet(1, 8, num_classes=5).to(DEV)
    ema = make_ema(net)
    opt = FusedAdamWEMA(net.parameters())
    net(torch.from_numpy(synth.image(1, 32)).to(DEV))['segmentation'].mean().backward()
    opt.fused_step(max_norm=1.0, ema_params=list(ema.parameters()), ema_alpha=0.0)
    path = str(tmp_path / 'fold_0_latest.pth')
    save_checkpoint(path, 3, net, ema, opt)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'model_state_dict', 'ema_model_state_dict', 'optimizer_state_dict'} and ck['epoch'] == 4
    net2 = UNet(1, 8, num_classes=5).to(DEV)
    assert load_checkpoint(path, net2, make_ema(net2), FusedAdamWEMA(net2.parameters())) == 4
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('C,shape', [(26, (24, 20, 32)), (42, (9, 7, 5)), (5, (16, 16, 16)), (8, (3, 3, 3))])
def test_unpack_bits_matches_numpy(C, shape):
    """rsuper_unpack_bits == np.unpackbits(np.packbits(x, axis=0), axis=0)[:C] (dataset_abdomenatlas_UFO.py:955,1031-1034), bit-exact."""
    import numpy as np
    from rsuper_amd.training.dataset import pack_bits, unpack_bits_device
    rng = np.random.default_rng(C)
    x = rng.random((2, C) + shape) < 0.3
    packed = pack_bits(x)
    assert packed.shape == (2, -(-C // 8)) + shape
    want = np.stack([np.unpackbits(packed[b], axis=0)[:C] for b in range(2)])
    got = unpack_bits_device(torch.from_numpy(packed).to(DEV), C).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, want) and np.array_equal(got.astype(bool), x)


@pytest.mark.gpu
def test_packed_batch_gives_same_loss():
    """calculate_loss on a batch ingested from bit-packed volumes == on the plain uint8 batch."""
    import argparse, sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import synth
    from rsuper_amd.training import losses_foundation as lf
    from rsuper_amd.training.dataset import pack_bits, ingest_packed_batch
    classes = synth.TINY_CLASSES
    B, S = 2, 32
    bt = synth.batch(B, S, classes, ['mask', 'report'], seed=11)
    logits = torch.from_numpy(synth.logits(B, len(classes), S, seed=5)).to(DEV)
    la = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                            ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                            classification_branch=False)

    def run(batch):
        r = lf.calculate_loss({'segmentation': logits.clone().requires_grad_(True)}, batch['label'], batch['unk_channels'], la, None, batch['mask'],
                              batch['volumes'], batch['diameters'], classes)
        return {k: float(v) for k, v in r.items()}
    plain = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
    packed = ingest_packed_batch({'label': pack_bits(bt['label']), 'unk_channels': pack_bits(bt['unk_channels']), 'mask': pack_bits(bt['mask']),
                                  'volumes': bt['volumes'], 'diameters': bt['diameters']}, len(classes), DEV)
    a, b = run(plain), run(packed)
    assert a.keys() == b.keys() and all(a[k] == b[k] for k in a), (a, b)
    # segmentation-only supervision: the label stays bit-packed and the loss kernels read the bits (csrc/loss.hip rsuper_plane_partials_fwd2 / _bwd2);
    # planes of the unknown map without a set voxel are skipped by flag.  Loss values AND the gradient of the logits must be bit-identical to the uint8 path.
    from rsuper_amd.training.dataset import PackedBits
    la0 = argparse.Namespace(**{**vars(la), 'report_volume_loss_basic': 0.0})
    kept = ingest_packed_batch({'label': pack_bits(bt['label']), 'unk_channels': pack_bits(bt['unk_channels']), 'mask': pack_bits(bt['mask']),
                                'volumes': bt['volumes'], 'diameters': bt['diameters']}, len(classes), DEV, keep_label_packed=True)
    assert isinstance(kept['label'], PackedBits) and kept['label']._u8 is None

    def run_g(batch):
        x = logits.clone().requires_grad_(True)
        r = lf.calculate_loss({'segmentation': x}, batch['label'], batch['unk_channels'], la0, None, batch['mask'], batch['volumes'], batch['diameters'], classes)
        r['overall'].backward()
        return {k: float(v) for k, v in r.items()}, x.grad
    (ra, ga), (rb, gb) = run_g(plain), run_g(kept)
    assert ra == rb, (ra, rb)
    assert torch.equal(ga, gb)
    assert kept['label']._u8 is None, 'the packed label must not have been inflated on the segmentation-only path'


@pytest.mark.gpu
@pytest.mark.parametrize('C,shape', [(5, (8, 8, 16)), (26, (16, 16, 16)), (42, (4, 12, 20)), (9, (3, 5, 7))])
def test_packed_plane_selection_and_flags_match_numpy(C, shape):
    """rsuper_unpack_bits_sel / rsuper_plane_any_bits (PackedBits.planes / class_flags / sample_any) against numpy on the unpacked volume, bit-exact:
    selected planes hold np.unpackbits(...)[c], planes neither selected nor flagged keep their previous bytes, flags == per-(sample, class) any()."""
    from rsuper_amd.training.dataset import pack_bits, PackedBits
    rng = np.random.default_rng(C)
    x = rng.random((3, C) + shape) < 0.02
    x[:, 1] = False                                  # an empty class everywhere
    x[1] = False                                     # an empty sample
    x[2, C - 1] = True
    packed = torch.from_numpy(pack_bits(x)).to(DEV)
    V = int(np.prod(shape))
    pb = PackedBits(packed, C)                       # (voxel counts that are no multiple of 16 take the element-wise fallback of class_flags / sample_any)
    fl = pb.class_flags().cpu().numpy().reshape(3, C)
    assert np.array_equal(fl.astype(bool), x.reshape(3, C, -1).any(-1))
    assert np.array_equal(pb.sample_any().cpu().numpy(), x.reshape(3, -1).any(-1))
    from rsuper_amd.hip import lib
    L = lib.lib()
    chs = [0, C - 1] if C > 8 else [C - 2]
    force = torch.zeros(C, dtype=torch.uint8); force[chs] = 1
    force = force.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    for use_flags in ((False, True) if V % 16 == 0 else (False,)):
        out = torch.full((3, C) + shape, 7, device=DEV, dtype=torch.uint8)
        flags = PackedBits(packed, C).class_flags() if use_flags else None
        assert L.rsuper_unpack_bits_sel(packed.data_ptr(), out.data_ptr(), 3, packed.shape[1], C, V, flags.data_ptr() if use_flags else None, force.data_ptr(), st) == 0
        got = out.cpu().numpy()
        written = np.zeros((3, C), bool)
        written[:, chs] = True
        if use_flags:
            written |= x.reshape(3, C, -1).any(-1)
        for b in range(3):
            for c in range(C):
                if written[b, c]:
                    assert np.array_equal(got[b, c].astype(bool), x[b, c]) and got[b, c].max() <= 1, (b, c)
                else:
                    assert (got[b, c] == 7).all(), (b, c)
    # both tables null == rsuper_unpack_bits
    out = torch.full((3, C) + shape, 7, device=DEV, dtype=torch.uint8)
    assert L.rsuper_unpack_bits_sel(packed.data_ptr(), out.data_ptr(), 3, packed.shape[1], C, V, None, None, st) == 0
    assert np.array_equal(out.cpu().numpy().astype(bool), x)


@pytest.mark.gpu
@pytest.mark.parametrize('loss,S', [('ball_dice_last', 32), ('ball_dice_both', 32), ('dice_volume', 32), ('ball_dice_both', 18), ('dice_volume', 18)])
def test_fully_packed_batch_gives_same_report_loss_and_gradient(loss, S):
    """SURVEY 8f-2 on the path R-Super trains: label, unknown map AND segment mask stay bit-packed under report supervision -- the
    segmentation term reads the label bits, the report terms get the lesion planes only (PackedBits.planes), the unknown map's plane flags come from the packed
    bytes.  Every loss key and the gradient of the logits are bit-identical to the uint8 batch (which the golden fixtures pin), no volume is inflated whole."""
    from rsuper_amd.training import losses_foundation as lf
    from rsuper_amd.training.dataset import PackedBits, pack_bits, ingest_packed_batch
    classes = synth.TINY_CLASSES
    B = 2                                            # S = 18: 5832 voxels per plane, not a multiple of 16 (ragged paths of every kernel involved)
    bt = synth.batch(B, S, classes, ['mask', 'report'], seed=11)
    logits = torch.from_numpy(synth.logits(B, len(classes), S, seed=5)).to(DEV)
    la = argparse.Namespace(loss=loss, aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                            ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                            classification_branch=False)
    plain = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
    kept = ingest_packed_batch({'label': pack_bits(bt['label']), 'unk_channels': pack_bits(bt['unk_channels']), 'mask': pack_bits(bt['mask']),
                                'volumes': bt['volumes'], 'diameters': bt['diameters']}, len(classes), DEV, keep_packed=True)
    assert all(isinstance(kept[k], PackedBits) for k in ('label', 'unk_channels', 'mask'))

    def run_g(batch, prefetch):
        x = logits.clone().requires_grad_(True)
        pre = lf.prepare_report_supervision(batch['label'], batch['unk_channels'], batch['mask'], batch['volumes'], batch['diameters'], classes, la) if prefetch else None
        r = lf.calculate_loss({'segmentation': x}, batch['label'], batch['unk_channels'], la, None, batch['mask'], batch['volumes'], batch['diameters'], classes, pre=pre)
        r['overall'].backward()
        return {k: float(v) for k, v in r.items()}, x.grad
    ra, ga = run_g(plain, False)
    for prefetch in (False, True):
        rb, gb = run_g(kept, prefetch)
        assert ra == rb, (prefetch, ra, rb)
        assert torch.equal(ga, gb), prefetch
    assert all(kept[k]._u8 is None for k in ('label', 'unk_channels', 'mask')), 'no packed volume may have been inflated whole'


@pytest.mark.gpu
@pytest.mark.parametrize('fill', ['dense', 'segment', 'empty'])
@pytest.mark.parametrize('d_odd', [5, 9, 21])
def test_ball_conv_two_stage_equals_direct(d_odd, fill):
    """(Consistency of two kernels of this repo, NOT a parity claim: the reference-pinned checks of the two-stage path are check_ball_large /
    test_report_loss_fullsize_matches_reference_and_oracle.)  The separable two-stage correlation gives the direct kernel's values (f32 summation-order tolerance) and argmax -- on a dense
    volume, on one that is zero outside an organ-like segment (the second stage then visits the occupied (z, y) rows only) and on an
    all-zero volume (first index wins)."""
    from rsuper_amd.hip import lib
    L = lib.lib()
    D, H, W = 24, 40, 28
    g = torch.Generator(device=DEV).manual_seed(d_odd)
    x = torch.rand((D, H, W), device=DEV, generator=g)
    if fill == 'segment':
        m = torch.zeros_like(x)
        m[5:17, 3:37:2, 4:20] = 1.0                      # every other row empty inside the box, rows on both sides of a 32-row word
        m[20, 33, 7] = 1.0
        x = x * m
    elif fill == 'empty':
        x = torch.zeros_like(x)
    outs, keys = [], []
    for two_stage in (False, True):
        best = torch.zeros(1, device=DEV, dtype=torch.int64)
        conv = torch.empty((D, H, W), device=DEV)
        ws = torch.empty((L.rsuper_ball_workspace_floats(D, H, W, d_odd),), device=DEV) if two_stage else None
        rc = L.rsuper_ball_conv_argmax(x.data_ptr(), D, H, W, d_odd, 1.5 * d_odd / 2.0, best.data_ptr(), conv.data_ptr(),
                                       ws.data_ptr() if ws is not None else None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(conv.cpu()); keys.append(int(best.item()) & 0xFFFFFFFF)
    assert torch.allclose(outs[0], outs[1], rtol=2e-5, atol=1e-5)
    assert keys[0] == keys[1]
    if fill == 'empty':
        assert 0xFFFFFFFF - keys[1] == 0 and float(outs[1].abs().max()) == 0.0


@pytest.mark.gpu
def test_grad_reducer_single_rank_nccl_matches_plain():
    """wrap_ddp's GradReducer on a 1-rank RCCL group: gradients land in the flat buckets, two optimiser steps give exactly
    the parameters of the un-wrapped run (mean over one rank = identity)."""
    import os
    import torch.distributed as dist
    from rsuper_amd.hip import ops
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import train_step, wrap_ddp, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    from rsuper_amd.training import losses_foundation as lf
    classes = synth.TINY_CLASSES
    B, S = 2, 32
    bt = synth.batch(B, S, classes, ['mask', 'mask'], seed=5)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
    batch['image'] = torch.from_numpy(synth.image(B, S, seed=3)).to(DEV)
    la = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                            ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                            classification_branch=False, ema=True, ema_alpha=0.99)

    def run(wrapped):
        torch.manual_seed(0)
        net = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='f32').to(DEV)
        ema = make_ema(net)
        model = wrap_ddp(net, 0) if wrapped else net
        opt = FusedAdamWEMA(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        for step in range(2):
            train_step(model, ema, opt, batch, la, classes, step)
        if wrapped:
            red = net._rsuper_reducer
            for p in net.parameters():
                b, off = red._slot[p]
                assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off
            red.remove()
        return [p.detach().clone() for p in net.parameters()]
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        a = run(True)
        os.environ['RSUPER_DDP_BF16'] = '1'                 # bf16 buckets on the wire: the same run with gradients rounded to bf16 on the way
        try:
            w = run(True)
        finally:
            del os.environ['RSUPER_DDP_BF16']
    finally:
        dist.destroy_process_group()
        ops.GRAD_DEST = None
    b = run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # two AdamW steps at lr 1e-3 from bf16-rounded gradients: parameters within a few 1e-4 of the exact run, and not identical to it
    assert max(float((x - y).abs().max()) for x, y in zip(w, b)) < 2e-3 and any(not torch.equal(x, y) for x, y in zip(w, b))


@pytest.mark.gpu
def test_conv_rejects_tensors_beyond_32bit_offsets():
    """Activation tensors of 4 GiB or more exceed the buffer-addressed staging's 32-bit byte offsets: the C ABI refuses them
    (RS_ERR_UNSUPPORTED) instead of wrapping around.  Only the size check runs; no memory of that size is touched."""
    from rsuper_amd.hip import lib
    L = lib.lib()
    d = torch.zeros(16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    N, D, H, W, C = 4, 256, 256, 256, 64                      # 67 M voxels x 128 B = 8.6 GB
    rc = L.rsuper_conv3_igemm(lib.BF16, 0, d.data_ptr(), C, C, None, None, 0, 0, None, d.data_ptr(), C, 64, N, D, H, W, d.data_ptr(), C, None, 0, None,
                              None, 0, 0, None, None, 0, 0, None, st)
    assert rc != 0
    rc = L.rsuper_conv3_wgrad(lib.BF16, 1, d.data_ptr(), C, C, None, None, 0, 0, None, d.data_ptr(), C, C, None, 0, 0, d.data_ptr(), None, d.data_ptr(),
                              N, D, H, W, 1, st)
    assert rc != 0


@pytest.mark.gpu
def test_training_converges_bf16_like_f32():
    """End-to-end sanity beyond the 2-step golden parity: 40 optimiser steps of the tiny UNet on one synthetic batch reduce
    the segmentation loss, and the bf16 production path tracks the f32 parity path."""
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    B, S = 2, 32
    bt = synth.batch(B, S, classes, ['mask', 'mask'], seed=9)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
    batch['image'] = torch.from_numpy(synth.image(B, S, seed=4)).to(DEV)
    la = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                            ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                            classification_branch=False, ema=True, ema_alpha=0.99)
    curves = {}
    for dt in ('f32', 'bf16'):
        torch.manual_seed(0)
        net = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=dt).to(DEV)
        ema = make_ema(net)
        opt = FusedAdamWEMA(net.parameters(), lr=2e-3, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        losses = []
        for step in range(40):
            la_out, gn = train_step(net, ema, opt, batch, la, classes, step)
            losses.append(float(la_out['overall'].detach()))
        assert all(np.isfinite(losses)), losses
        curves[dt] = losses
    for dt, c in curves.items():
        assert c[-1] < 0.75 * c[0], (dt, c[0], c[-1])
    assert abs(curves['bf16'][-1] - curves['f32'][-1]) < 0.1 * curves['f32'][0], (curves['bf16'][-1], curves['f32'][-1])


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['distinct', 'ties', 'masked', 'k_all'])
def test_topk_select_device_matches_oracle(case):
    """rsuper_topk_select (device-resident radix select) == oracle canonical top-k (value desc, index asc), bit-exact,
    including threshold ties that must be split in index order."""
    from rsuper_amd.hip import lib
    from oracle import morph as om
    L = lib.lib()
    rng = np.random.default_rng(7)
    V = 24 * 20 * 28
    x = rng.random(V).astype(np.float32)
    m = np.ones(V, np.uint8)
    k = 1500
    if case == 'ties':
        x = (rng.integers(0, 40, V) / 40.0).astype(np.float32)       # many equal values: the threshold class is split by index
    elif case == 'masked':
        m = (rng.random(V) < 0.5).astype(np.uint8)
    elif case == 'k_all':
        k = V
    want = om.topk_mask(np.ascontiguousarray((x * m).astype(np.float32)), k)
    xd, md = torch.from_numpy(x).to(DEV), torch.from_numpy(m).to(DEV)
    out = torch.empty(V, device=DEV, dtype=torch.uint8)
    ws = torch.empty(260, device=DEV, dtype=torch.int32)
    rc = L.rsuper_topk_select(xd.data_ptr(), md.data_ptr(), V, k, out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = out.cpu().numpy()
    assert got.sum() == k and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['plain', 'ties', 'sparse_positive'])
def test_topk_select_multi_equals_three_selections_and_mask(case):
    """rsuper_topk_select_multi (three counts over one (x, ball) in one pass sequence, clipped to the ball) == three oracle top-k masks
    of x * ball AND-ed with the ball, bit-exact -- incl. the case where k exceeds the number of positive values (threshold 0: the dense
    top-k then picks zeros by index, inside and outside the ball; only the inside ones survive the clip)."""
    import ctypes
    from rsuper_amd.hip import lib
    from oracle import morph as om
    L = lib.lib()
    rng = np.random.default_rng(11)
    V = 24 * 20 * 28
    x = rng.random(V).astype(np.float32)
    if case == 'ties':
        x = (rng.integers(0, 30, V) / 30.0).astype(np.float32)
    ball = (rng.random(V) < 0.3).astype(np.uint8)
    if case == 'sparse_positive':
        x = x * (rng.random(V) < 0.05)                            # ~200 positive values inside the ball, k up to 3000
    ks = (1500, 1200, 3000)
    xm = np.ascontiguousarray((x * ball).astype(np.float32))
    want = [om.topk_mask(xm, k) & ball for k in ks]
    xd, bd = torch.from_numpy(x.astype(np.float32)).to(DEV), torch.from_numpy(ball).to(DEV)
    out = torch.empty((3, V), device=DEV, dtype=torch.uint8)
    ws = torch.empty(3 * 260, device=DEV, dtype=torch.int32)
    rc = L.rsuper_topk_select_multi(xd.data_ptr(), bd.data_ptr(), V, (ctypes.c_uint * 3)(*ks), 3, out.data_ptr(), ws.data_ptr(), 1,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = out.cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], want[i]), (case, i, int(got[i].sum()), int(want[i].sum()))


@pytest.mark.gpu
def test_plane_any_matches_torch():
    from rsuper_amd.training.losses_foundation import _plane_any
    g = torch.Generator(device=DEV).manual_seed(3)
    t = (torch.rand((3, 5, 8, 12, 16), device=DEV, generator=g) < 0.0005).to(torch.uint8)
    t[1] = 0
    assert torch.equal(_plane_any(t, 2), t.flatten(2).any(2)) and torch.equal(_plane_any(t, 1), t.flatten(1).any(1))
    odd = (torch.rand((2, 7, 5, 3), device=DEV, generator=g) < 0.01).to(torch.uint8)      # V % 16 != 0 -> ATen path
    assert torch.equal(_plane_any(odd, 1), odd.flatten(1).any(1))
    one = torch.zeros((1, 1, 5, 7, 3), device=DEV, dtype=torch.uint8); one[0, 0, 4, 6, 2] = 1
    assert bool(_plane_any(one, 0))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_gloo():
    """The N = 2 flow of bench.py as the driver launches it (torchrun, one process per rank, GradReducer, barrier + max over
    ranks, rank-0 JSON) -- on a single-GPU box the two ranks share the device through RSUPER_DIST_BACKEND=gloo (RCCL refuses
    duplicate devices); only RCCL's own transport is not exercised."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSUPER_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29571', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '64',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['scaling'] == 'weak'
    assert d['value'] > 0 and np.isfinite(d['config']['final_loss'])


def test_two_ranks_match_two_reference_ranks_averaged():
    """N-GPU parity = N independent reference ranks averaged.  Two ranks (one process each, sharing this GPU through gloo) run
    the HIP forward / loss / backward of the tiny UNet on their own batches under wrap_ddp; after GradReducer.finish() every parameter gradient equals
    the mean of the unmodified reference's two ranks (tests/golden/ddp2.npz) and each rank's loss values equal its reference rank's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSUPER_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29575', os.path.join(root, 'tests', 'ddp_fixture_gpu_worker.py')], env=env, capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert 'DDP_FIXTURE_GPU_OK' in r.stdout


@pytest.mark.gpu
def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with NO launcher environment (how the driver ran N = 1 in round 1): bench.py re-executes
    itself under torch.distributed.run, one rank per GPU; on this single-GPU box the ranks share the device through gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['RSUPER_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '64',
                        '--roofline-steps', '1', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['roofline']['frac'] > 0


@pytest.mark.gpu
def test_train_epoch_driver_runs_reference_loop(tmp_path):
    """train_net / train_epoch (train_ddp.py:65-389) end to end on a synthetic dataset: ChunkedSampler + DataLoader batches with
    the reference's keys, the epoch length (iter_per_epoch batches from the sampler), meters over every loss key, LR schedule per epoch,
    `latest` checkpoint per epoch, and --resume continuing from it."""
    import logging
    import os
    from rsuper_amd.train_ddp import get_parser, main_worker
    from rsuper_amd.training.dataset import SyntheticUFODataset
    classes = ['kidney_left', 'kidney_right', 'liver', 'pancreas', 'pancreatic_lesion']
    ds = SyntheticUFODataset(classes, size=32, length=12, seed=3)
    b = ds[1]
    assert b['label'].shape == (5, 32, 32, 32) and b['image'].shape == (1, 32, 32, 32) and b['volumes'].sum() > 0 and b['mask'].any()
    argv = ['--epochs', '2', '--batch_size', '2', '--cp_path', str(tmp_path) + '/', '--unique_name', 'drv', '--loss', 'ball_dice_both',
            '--report_volume_loss_basic', '0.1']
    args = get_parser(argv)
    args.base_chan, args.iter_per_epoch, args.print_freq, args.compute_dtype = 8, 2, 1, 'f32'
    seen = []

    class Count(logging.Handler):
        def emit(self, rec):
            if 'epoch: [' in rec.getMessage():
                seen.append(rec.getMessage())
    h = Count()
    logging.getLogger().addHandler(h); logging.getLogger().setLevel(logging.INFO)
    try:
        hist = main_worker(0, 1, 0, args, trainset=ds)
    finally:
        logging.getLogger().removeHandler(h)
    assert len(hist) == 2 and all(np.isfinite(v) for e in hist for v in e.values())
    assert {'segmentation', 'overall'} <= set(hist[0]) and hist[1]['overall'] < hist[0]['overall'] * 1.5
    # ChunkedSampler serves iter_per_epoch * batch_size samples per epoch (train_ddp.py:106-112), so the loader is exhausted after
    # iter_per_epoch batches and the `> iter_per_epoch` break (:380-383) never fires: two progress lines per epoch
    assert len(seen) == 2 * 2, seen
    ck = os.path.join(str(tmp_path), 'abdomenatlas_ufo', 'drv', 'fold_0_latest.pth')
    assert os.path.exists(ck)
    args2 = get_parser(argv[:1] + ['3'] + argv[2:] + ['--resume'])
    args2.base_chan, args2.iter_per_epoch, args2.print_freq, args2.compute_dtype = 8, 2, 1, 'f32'
    hist2 = main_worker(0, 1, 0, args2, trainset=ds)
    assert len(hist2) == 1 and args2.start_epoch == 2   # resumed after the two finished epochs


@pytest.mark.gpu
def test_augmented_loader_feeds_packed_ingest(tmp_path):
    """Crop directory -> AugmentedCropDataset(packed=True) -> DataLoader -> ingest_packed_batch on the device gives exactly the
    volumes the host-unpacking loader (reference behaviour, dataset_abdomenatlas_UFO.py:1028-1036,1083-1101) yields, and one epoch of
    the driver runs from the directory through `main(['--load_augmented', ...])`'s dataset path."""
    import os
    import yaml
    import synth
    from rsuper_amd.training.dataset import AugmentedCropDataset, save_crop, ingest_packed_batch, SyntheticUFODataset
    from rsuper_amd.train_ddp import get_parser, main_worker, load_label_names
    classes = ['kidney_left', 'kidney_right', 'liver', 'pancreas', 'pancreatic_lesion']
    src = SyntheticUFODataset(classes, size=32, length=6, seed=5)
    crops, root = str(tmp_path / 'crops'), str(tmp_path / 'root')
    os.makedirs(os.path.join(root, 'list'))
    with open(os.path.join(root, 'list', 'label_names.yaml'), 'w') as f:
        yaml.safe_dump(list(reversed(classes)), f)
    for i in range(len(src)):
        s = src[i]
        name = 'BDMAP_%08d' % i
        report = i % 2 == 1
        rows = [{'Standardized Organ': 'pancreas', 'Standardized Location': 'pancreas head', 'Tumor Size (mm)': '%g' % d[0].item()}
                for d in s['diameters'] if d[0] > 0]
        save_crop(crops, name + '.npy', name + '_gt.npy', s['image'], s['label'],
                  s['unk_channels'] if report else None, s['mask'] if report else None,
                  {'tumor_in_crop': 'pancreas'} if report else None, rows if report else None)
    args = get_parser(['--epochs', '1', '--batch_size', '2', '--cp_path', str(tmp_path) + '/', '--unique_name', 'aug',
                       '--load_augmented', '--save_destination', crops, '--data_root', root, '--loss', 'ball_dice_both',
                       '--report_volume_loss_basic', '0.1'])
    assert load_label_names(args) == classes
    host = AugmentedCropDataset.from_directory(crops, classes, packed=False, augment=False)
    dev = AugmentedCropDataset.from_directory(crops, classes, packed=True, augment=False)
    assert len(host) == 6 and len(host.ufo_paths) == 3
    for bi, (hb, db) in enumerate(zip(torch.utils.data.DataLoader(host, batch_size=3), torch.utils.data.DataLoader(dev, batch_size=3))):
        got = ingest_packed_batch(db, len(classes))
        for k in ('label', 'unk_channels', 'mask'):
            assert got[k].dtype == torch.uint8 and torch.equal(got[k].cpu(), hb[k].to(torch.uint8)), k
        for k in ('image', 'volumes', 'diameters'):
            assert torch.equal(got[k].cpu(), hb[k].float()), k
        for j in range(3):                                               # odd crops carry report supervision
            assert bool(got['volumes'][j].sum() > 0) == bool((3 * bi + j) % 2) == bool(got['mask'][j].any())
    args.base_chan, args.iter_per_epoch, args.print_freq, args.compute_dtype = 8, 2, 1, 'f32'
    np.random.seed(0); torch.manual_seed(0)
    train = AugmentedCropDataset.from_directory(crops, classes, packed=True)
    hist = main_worker(0, 1, 0, args, trainset=train)
    assert len(hist) == 1 and all(np.isfinite(v) for v in hist[0].values()) and 'overall' in hist[0]


@pytest.mark.gpu
def test_gemm_library_switch_is_scoped():
    """medformer_utils.gemm_library: hipBLASLt for the calls inside the block when the reduction is long and MedFormer's rocBLAS default is
    active, the default restored afterwards (also on an exception); inactive -> never touches the process setting."""
    from rsuper_amd.model.dim3 import medformer_utils as mu
    get = torch.backends.cuda.preferred_blas_library
    before, was_active = get(), mu.gemm_library.active
    try:
        mu.gemm_library.active = True
        get('cublas')
        with mu.gemm_library(mu.LT_MIN_K):
            assert 'lt' in str(get()).lower()
            y = torch.mm(torch.ones(8, mu.LT_MIN_K, device='cuda'), torch.ones(mu.LT_MIN_K, 8, device='cuda'))
        assert 'lt' not in str(get()).lower() and float(y[0, 0]) == mu.LT_MIN_K
        with mu.gemm_library(mu.LT_MIN_K - 1):
            assert 'lt' not in str(get()).lower()
        with pytest.raises(ZeroDivisionError):
            with mu.gemm_library(4 * mu.LT_MIN_K):
                1 / 0
        assert 'lt' not in str(get()).lower()
        mu.gemm_library.active = False
        get('cublaslt')
        with mu.gemm_library(4 * mu.LT_MIN_K):
            pass
        assert 'lt' in str(get()).lower()
    finally:
        mu.gemm_library.active = was_active
        get(before)


@pytest.mark.gpu
def test_medformer_shipped_widths_match_oracle_f32():
    """The SHIPPED MedFormer widths (config/abdomenatlas_ufo/medformer_3d.yaml: base 32, channels 64..320, 4 / 8 / 10 heads of 32, 27 map
    tokens, fusion depth 2) -- the shapes the fused attention core is built for, which the tiny fixture (8 tokens, heads of 16) does not
    reach -- on a 32^3 crop in f32 against the CPU restatement oracle/medformer_oracle.py with the same seeded state_dict: both heads and
    the gradients of every parameter.  These nets are ill-conditioned in fp32 (InstanceNorm over 2^3 .. 8^3 voxels, 18 attention blocks:
    the fp32 oracle itself is 1.5e-4 on the logits and 0.36 on its worst gradient tensor away from its float64 evaluation), so the bound
    is relative to that noise: the HIP path may be at most 3x as far from the float64 oracle as the fp32 oracle is (measured 1.6x on the
    logits, 1.0x on the gradients)."""
    import synth
    from oracle import medformer_oracle as mo
    from rsuper_amd.model.dim3.medformer import MedFormer
    cfg = dict(base_chan=32, map_size=[3, 3, 3], conv_num=[2, 0, 0, 0, 0, 0, 2, 2], trans_num=[0, 2, 4, 6, 4, 2, 0, 0],
               num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4, aux_loss=True)
    ncls = 5
    net = MedFormer(1, ncls, compute_dtype='f32', **cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd_np = synth.fill_state_dict(shapes, 23)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    net = net.to('cuda')
    x = torch.from_numpy(synth.image(1, 32, seed=77))
    go = torch.from_numpy(synth.rng(5).standard_normal((1, ncls, 32, 32, 32)).astype(np.float32)) / 32 ** 3
    ga = torch.from_numpy(synth.rng(6).standard_normal((1, ncls, 32, 32, 32)).astype(np.float32)) / 32 ** 3
    res = {}
    for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
        sd = {k: torch.from_numpy(v).to(dt if v.dtype == np.float32 else None).clone().requires_grad_(v.dtype == np.float32) for k, v in sd_np.items()}
        yr, ar = mo.medformer_forward(sd, x.to(dt), cfg)
        ((yr * go.to(dt)).sum() + (ar * ga.to(dt)).sum()).backward()
        res[name] = (yr.detach().double(), ar.detach().double(), {k: v.grad.double() for k, v in sd.items() if v.grad is not None})
    y, a = net(x.to('cuda'))['segmentation']
    ((y * go.to('cuda')).sum() + (a * ga.to('cuda')).sum()).backward()
    torch.cuda.synchronize()
    hip = (y.detach().cpu().double(), a.detach().cpu().double(), {k: p.grad.cpu().double() for k, p in net.named_parameters()})
    ref = res['f64']
    assert set(hip[2]) == set(ref[2])
    rel = lambda g, r: float((g - r).abs().max() / r.abs().max())
    gmax = max(float(v.abs().max()) for v in ref[2].values())
    worst = lambda t: max(float((t[2][k] - ref[2][k]).abs().max()) / max(float(ref[2][k].abs().max()), 1e-3 * gmax) for k in ref[2])
    for i, what in ((0, 'logits'), (1, 'aux logits')):
        e_hip, e_f32 = rel(hip[i], ref[i]), rel(res['f32'][i], ref[i])
        assert e_hip <= 3 * e_f32 + 1e-5, (what, e_hip, e_f32)
    assert worst(hip) <= 3 * worst(res['f32']) + 1e-3, (worst(hip), worst(res['f32']))


@pytest.mark.gpu
def test_medformer_fused_attention_matches_aten_composition():
    """The HIP attention core (csrc/battn.hip) against the ATen composition it replaces (einsum / soft-max / head rearranges), through
    the whole tiny MedFormer in f32: logits and every parameter gradient.  Same mathematics in a different summation order on an
    ill-conditioned tiny net: 2e-3 of the largest gradient."""
    import synth
    from rsuper_amd.model.dim3.medformer import MedFormer
    from rsuper_amd.model.dim3 import medformer_utils as mu
    cfg = {k: v for k, v in synth.MEDFORMER_TINY.items() if k not in ('size', 'seed')}
    x = torch.from_numpy(synth.image(1, 32, seed=3)).to('cuda')

    def run(fused):
        old = mu.FUSED_ATTENTION
        mu.FUSED_ATTENTION = fused
        try:
            torch.manual_seed(0)
            net = MedFormer(1, len(synth.TINY_CLASSES), compute_dtype='f32', **cfg).to('cuda')
            y, aux = net(x)['segmentation']
            (y.square().mean() + aux.square().mean()).backward()
            return y.detach(), aux.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        finally:
            mu.FUSED_ATTENTION = old

    y1, a1, g1 = run(True)
    y0, a0, g0 = run(False)
    assert float((y1 - y0).abs().max() / y0.abs().max()) < 1e-4 and float((a1 - a0).abs().max() / a0.abs().max()) < 1e-4
    gmax = max(float(g.abs().max()) for g in g0.values())
    worst = max(float((g1[k] - g0[k]).abs().max()) for k in g0) / gmax
    assert worst < 2e-3, worst


@pytest.mark.gpu
@pytest.mark.parametrize('seed,B,L,use_vol,apply_dice,std_ce,weighted', [(1, 2, 1, True, True, False, False), (2, 3, 2, True, True, False, True),
                                                                          (3, 2, 1, False, True, True, True), (4, 4, 3, True, False, False, True),
                                                                          (5, 1, 1, True, True, False, False)])
def test_report_from_sums_kernel_matches_torch_autograd(seed, B, L, use_vol, apply_dice, std_ce, weighted):
    """csrc/loss.hip report_from_sums_kernel (volume loss + ball-loss tail from the per-plane sums, with Jacobians) against the same algebra written in
    torch with autograd in float64 (losses_foundation.py:250-395, :1625-1661, :1793-1811, :541-607): values and d / d sums, incl. clamp edges
    (volumes inside the tolerance band, alpha at its clamps) and plans of both kinds."""
    from rsuper_amd.training import losses_foundation as lf
    g = torch.Generator().manual_seed(seed)
    V = 32 ** 3
    kinds = [int(torch.randint(0, 2, (1,), generator=g)) for _ in range(B)]
    kinds[0] = 1
    if B > 1:
        kinds[1] = 0
    R = (L * B if use_vol else 0) + sum(L if k == 0 else 1 for k in kinds)
    sums = torch.rand((R, 6), generator=g, dtype=torch.float64)
    sums[:, 0] *= 5000.0                                   # S: summed BCE
    sums[:, 1] = sums[:, 1] * 3000.0 + 10.0                # A = sum P
    sums[:, 2] = sums[:, 2] * sums[:, 1].clamp(max=900.0)  # Bs = sum P T <= A, Cn
    sums[:, 3] = sums[:, 2] + torch.rand(R, generator=g, dtype=torch.float64) * 800.0
    sums[:, 4:] *= 4000.0
    if R > 2:
        sums[2, 1], sums[2, 2], sums[2, 3] = 50.0, 49.0, 2000.0        # FP << FN: alpha at the lower clamp
    flags = (torch.rand((B, 2 * L), generator=g) < 0.5).float()
    flags[:, L:] = 1.0
    flags[0, :L] = 0.0
    rvol = torch.rand(B, generator=g) * 3000.0 + 50.0
    if use_vol:
        rvol[0] = float(sums[0, 1]) * 1.05                 # inside the tolerance band: clamped to 0
    roww = (torch.rand(R, generator=g) + 0.5) if weighted else torch.ones(R)
    plan, row = [], L * B if use_vol else 0
    for k in kinds:
        plan += [k, row]
        row += L if k == 0 else 1
    tol = 0.2

    def torch_algebra(sm):
        out = {}
        if use_vol:
            x = torch.stack([sm[li * B:(li + 1) * B, 1] for li in range(L)], 1) * (1 - flags[:, :L].double())
            y = rvol.double()[:, None].expand(B, L) * flags[:, L:].double()
            loss = torch.abs(x - y) / (x + y + 500)
            v = torch.max((1 - tol) * y, y.clamp(max=100))
            loss = torch.clamp(loss - torch.abs(v - y) / (v + y + 500), min=0, max=1)
            w = torch.stack([roww[li * B:(li + 1) * B] for li in range(L)], 1).double()
            out['vol'] = (loss * w).mean()
        lb, ld = [], []
        for q in range(len(kinds)):
            k, r0 = plan[2 * q], plan[2 * q + 1]
            rows = sm[r0:r0 + (L if k == 0 else 1)]
            w = roww[r0:r0 + rows.shape[0]].double()
            if k == 0:
                lb.append((rows[:, 0] * w).sum() / float(L * V))
            else:
                lb.append((rows[0, 0] if std_ce else rows[0, 4] + rows[0, 5]) / float(V) * w[0])
            TP, FP, FN = rows[:, 2], rows[:, 1] - rows[:, 2], rows[:, 3] - rows[:, 2]
            al = (FP / (FP + FN + 1e-5)).clamp(0.2, 0.8)
            ld.append(((1 - TP / (TP + al * FP + (1 - al) * FN + 1e-5)) * w).mean())
        out['bce'] = torch.stack(lb).mean()
        out['dice'] = torch.stack(ld).mean() if apply_dice else torch.zeros(())
        return out

    sm = sums.clone().requires_grad_(True)
    ref = torch_algebra(sm)
    dev = sums.float().to('cuda').requires_grad_(True)
    lb, ld, lv = lf._ReportFromSums.apply(dev, roww.to('cuda'), torch.tensor(plan, dtype=torch.int32, device='cuda'),
                                          flags.to('cuda') if use_vol else None, rvol.to('cuda') if use_vol else None, B, L, V, use_vol, tol,
                                          len(kinds), apply_dice, std_ce)
    for name, got in (('bce', lb), ('dice', ld)) + ((('vol', lv),) if use_vol else ()):
        assert abs(float(got) - float(ref[name])) <= 2e-6 * max(1.0, abs(float(ref[name]))), (name, float(got), float(ref[name]))
        if ref[name].requires_grad:
            gr, = torch.autograd.grad(ref[name], sm, retain_graph=True)
            gd, = torch.autograd.grad(got, dev, retain_graph=True)
            assert float((gd.cpu().double() - gr).abs().max()) <= 2e-6 * max(float(gr.abs().max()), 1e-12), name


@pytest.mark.gpu
def test_medformer_train_steps_deep_supervision():
    """MedFormer (8f-1) through the training step: deep-supervision output [final, aux] (medformer.py:205-222) into calculate_loss with
    aux_weight (losses_foundation.py:905-930), backward through the HIP conv stages and the attention stages, clip + fused AdamW + EMA over
    every parameter kind (conv, depthwise, Linear, LayerNorm, biases); finite, decreasing on a repeated batch, and reproducible."""
    import argparse
    import synth
    from rsuper_amd.model.dim3.medformer import MedFormer
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    cfg = {k: v for k, v in synth.MEDFORMER_TINY.items() if k not in ('size', 'seed')}
    largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                               ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                               classification_branch=False, ema=True, ema_alpha=0.99)
    bt = synth.batch(2, 32, classes, ['mask', 'report'], seed=11, diam_range=(4.0, 8.0), max_tumors=2)
    dev = 'cuda'
    batch = dict(image=torch.from_numpy(synth.image(2, 32, seed=5)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
                 unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
                 volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev))

    def run(mode):
        torch.manual_seed(0)
        net = MedFormer(1, len(classes), compute_dtype=mode, **cfg).to(dev)
        ema = make_ema(net)
        opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        hist = []
        for i in range(8):
            loss, gnorm = train_step(net, ema, opt, batch, largs, classes, i)
            hist.append({k: float(v.detach()) for k, v in loss.items()})
        return hist, net, ema

    for mode in ('f32', 'bf16'):
        h1, net, ema = run(mode)
        assert all(np.isfinite(v) for e in h1 for v in e.values()), h1
        assert {'segmentation', 'overall'} <= set(h1[0]) and h1[-1]['overall'] < h1[0]['overall'], (mode, h1[0], h1[-1])
        h2, _, _ = run(mode)
        assert h1 == h2, 'MedFormer training steps are not reproducible'
        moved = sum(float((a - b).abs().sum()) for a, b in zip(net.state_dict().values(), ema.state_dict().values()))
        assert moved > 0                                    # EMA lags the weights
        sd = net.state_dict()
        assert 'down2.trans_blocks.blocks.0.attn.feat_qv.depthwise.weight' in sd and 'map_fusion.fusion.layers.0.0.fn.to_qkv.weight' in sd


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['unet', 'medformer'])
def test_graphed_step_matches_eager(model):
    """rsuper_amd.graph.GraphedTrainStep: the training step captured in a hipGraph and replayed gives exactly the losses and parameters of
    the eager train_step, including a learning-rate change between replays (the lr, the Adam bias corrections and the EMA alpha are
    read from device memory), different batches (static input buffers) and the EMA ramp."""
    import argparse
    import synth
    from rsuper_amd.graph import GraphedTrainStep
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                               ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                               classification_branch=False, ema=True, ema_alpha=0.99)
    dev = 'cuda'
    batches = []
    for i in range(2):
        bt = synth.batch(2, 32, classes, ['mask', 'mask'], seed=11 + i)
        batches.append(dict(image=torch.from_numpy(synth.image(2, 32, seed=5 + i)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
                            unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
                            volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev)))

    def build():
        torch.manual_seed(0)
        if model == 'unet':
            from rsuper_amd.model.dim3.unet import UNet
            net = UNet(1, 8, num_classes=len(classes), compute_dtype='bf16').to(dev)
        else:
            from rsuper_amd.model.dim3.medformer import MedFormer
            net = MedFormer(1, len(classes), compute_dtype='bf16', **{k: v for k, v in synth.MEDFORMER_TINY.items() if k not in ('size', 'seed')}).to(dev)
        ema = make_ema(net)
        return net, ema, FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)

    def run(graphed):
        net, ema, opt = build()
        stepper = GraphedTrainStep(net, ema, opt, largs, classes, warmup=2) if graphed else None
        hist = []
        for i in range(16):                                           # 14 replays: past the 12th, where a captured ATen reduction once went
            if i == 4:                                                # wrong (the step verifies itself against an eager step at replays 1 and 12)
                opt.param_groups[0]['lr'] = 2e-4                      # epoch boundary of the LR schedule
            b = batches[i % 2]
            loss, gn = stepper(b, i) if graphed else train_step(net, ema, opt, b, largs, classes, i)
            hist.append((float(loss['overall'].detach()), float(gn)))
        if graphed:
            assert stepper.graph is not None and stepper.calls == 16 and stepper._replays == 14
        return hist, [p.detach().clone() for p in net.parameters()], [p.detach().clone() for p in ema.parameters()], opt

    h_e, p_e, e_e, opt_e = run(False)
    h_g, p_g, e_g, opt_g = run(True)
    assert h_e == h_g, (h_e, h_g)
    assert all(torch.equal(a, b) for a, b in zip(p_e, p_g)) and all(torch.equal(a, b) for a, b in zip(e_e, e_g))
    st_e, st_g = next(iter(opt_e.state.values())), next(iter(opt_g.state.values()))
    assert int(st_e['step']) == int(st_g['step']) == 16
    if model == 'medformer':
        # captures never record rocBLAS launches (rsuper_amd.graph._capture_safe_blas: the root cause of round 2's "garbage after a few
        # replays"): the process is on hipBLASLt and MedFormer's per-call library switch is off after a capture
        from rsuper_amd.model.dim3 import medformer_utils as mu
        assert 'lt' in str(torch.backends.cuda.preferred_blas_library()).lower() and mu.gemm_library.active is False
    with pytest.raises(ValueError):
        GraphedTrainStep(*build(), argparse.Namespace(**{**vars(largs), 'report_volume_loss_basic': 0.1}), classes)
    with pytest.raises(ValueError):                                   # the optimiser state must exist before the capture
        GraphedTrainStep(*build(), largs, classes, warmup=0)
    if model == 'unet':
        # self-verification: a replay whose gradient buffers disagree with the eager step is reported (here: scribbled on before the check)
        net, ema, opt = build()
        stepper = GraphedTrainStep(net, ema, opt, largs, classes, warmup=1)
        stepper(batches[0], 0)
        stepper(batches[0], 1)                                        # capture + replay 1 (verified, healthy)
        real = stepper._verify
        def broken(before):
            next(p for p in net.parameters() if p.grad is not None).grad.add_(1.0)
            return real(before)
        stepper._verify, stepper.verify_at = broken, (2,)
        with pytest.raises(RuntimeError, match='disagrees with the eager step'):
            stepper(batches[1], 2)
        # the exception's traceback keeps the stepper (and its hipGraph) in a reference cycle: collect it here, not at some later
        # point of another test's graph replays
        del stepper, real, broken, net, ema, opt
        import gc
        gc.collect()
        torch.cuda.synchronize()


@pytest.mark.gpu
def test_graphed_network_self_verification_reports_a_wrong_replay():
    """GraphedNetwork recomputes the gradients eagerly at replays 1, 12, 50, ... and compares them with the backward graph's static buffers.
    A healthy run passes silently (every other graph test runs with the check on); here the backward graph is wrapped so that one static
    gradient buffer is scribbled on after the replay -- what a defective replay looks like from the outside: RuntimeError naming the
    gradient.  With RSUPER_GRAPH_VERIFY=0 semantics (verify_at = ()) the same sequence goes through."""
    import synth
    from rsuper_amd.graph import GraphedNetwork
    from rsuper_amd.model.dim3.unet import UNet
    classes = synth.TINY_CLASSES
    img = torch.from_numpy(synth.image(1, 32, seed=3)).to('cuda')
    for check in (True, False):
        torch.manual_seed(0)
        net = UNet(1, 8, num_classes=len(classes), compute_dtype='f32').to('cuda')
        g = GraphedNetwork(net, warmup=1)
        if not check:
            g.verify_at, g.verify_every = (), 0
        y = g(img)['segmentation']

        class Scribbling:
            def __init__(self, graph):
                self.graph = graph

            def replay(self):
                self.graph.replay()
                g.static_grads[-1].add_(1.0)           # outc.bias

        g.bwd_graph = Scribbling(g.bwd_graph)
        if check:
            with pytest.raises(RuntimeError, match='disagrees with the eager step'):
                y.square().mean().backward()
        else:
            y.square().mean().backward()
            assert net.outc.weight.grad is not None


@pytest.mark.gpu
def test_graphed_step_without_per_step_sync_matches_eager():
    """GraphedTrainStep driven the way a training loop drives it -- no host read between steps, so the host runs many replays ahead of the
    GPU -- ends with exactly the parameters of the eager run.  Regression test: the step-dependent optimiser scalars (Adam bias
    corrections, EMA alpha) used to be staged through ONE reused pinned buffer, which a host running ahead overwrote before the
    asynchronous copy of the previous step had executed."""
    import argparse
    import synth
    from rsuper_amd.graph import GraphedTrainStep
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                               ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                               classification_branch=False, ema=True, ema_alpha=0.99)
    S = 64                                                    # ~ms of GPU work per step against ~0.1 ms of host work per replay
    bt = synth.batch(2, S, classes, ['mask', 'mask'], seed=41)
    batch = dict(image=torch.from_numpy(synth.image(2, S, seed=17)).to('cuda'), label=torch.from_numpy(bt['label']).to('cuda'),
                 unk_channels=torch.from_numpy(bt['unk_channels']).to('cuda'), mask=torch.from_numpy(bt['mask']).to('cuda'),
                 volumes=torch.from_numpy(bt['volumes']).to('cuda'), diameters=torch.from_numpy(bt['diameters']).to('cuda'))

    def run(graphed):
        torch.manual_seed(0)
        net = UNet(1, 16, num_classes=len(classes), compute_dtype='bf16').to('cuda')
        ema = make_ema(net)
        opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        st = GraphedTrainStep(net, ema, opt, largs, classes, warmup=2) if graphed else None
        for i in range(40):
            st(batch, i) if graphed else train_step(net, ema, opt, batch, largs, classes, i)      # nothing read back: the host runs ahead
        torch.cuda.synchronize()
        return [p.detach().clone() for p in net.parameters()], [p.detach().clone() for p in ema.parameters()]

    (p_e, e_e), (p_g, e_g) = run(False), run(True)
    assert all(torch.equal(a, b) for a, b in zip(p_e, p_g)) and all(torch.equal(a, b) for a, b in zip(e_e, e_g))


@pytest.mark.gpu
def test_graphed_network_many_replays_match_eager():
    """More than a dozen replays of the forward / backward graphs of a MedFormer whose deep-supervision head is tall enough (8192 rows)
    for the GEMM form of its bias gradient: losses and gradient norms equal to the eager run at EVERY step.  Regression test for a defect
    of captured ATen multi-block reductions on this ROCm -- `dy.sum(0)` over 27648 rows returned garbage from the 12th replay of a hipGraph
    on (eager and the first 11 replays were right), found by comparing an eager and a replayed full-size run gradient by gradient; the tall
    bias gradients are now a 1 x rows GEMM (model/dim3/medformer_utils.py _LinearFn)."""
    import argparse
    import synth
    from rsuper_amd.graph import GraphedNetwork
    from rsuper_amd.model.dim3.medformer import MedFormer
    from rsuper_amd.model.dim3 import medformer_utils as mu
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    largs = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                               ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                               classification_branch=False, ema=True, ema_alpha=0.99)
    S = 64
    bt = synth.batch(2, S, classes, ['mask', 'mask'], seed=31)
    batch = dict(image=torch.from_numpy(synth.image(2, S, seed=9)).to('cuda'), label=torch.from_numpy(bt['label']).to('cuda'),
                 unk_channels=torch.from_numpy(bt['unk_channels']).to('cuda'), mask=torch.from_numpy(bt['mask']).to('cuda'),
                 volumes=torch.from_numpy(bt['volumes']).to('cuda'), diameters=torch.from_numpy(bt['diameters']).to('cuda'))
    assert 2 * (S // 4) ** 3 >= 4096 >= 1 and mu.SPLITK_MIN_ROWS <= 2 * (S // 4) ** 3      # the aux head takes _LinearFn with the GEMM bias gradient

    def run(graphed):
        torch.manual_seed(0)
        net = MedFormer(1, len(classes), compute_dtype='bf16', **{k: v for k, v in synth.MEDFORMER_TINY.items() if k not in ('size', 'seed')}).to('cuda')
        ema = make_ema(net)
        opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        f = GraphedNetwork(net, warmup=2) if graphed else net
        hist = []
        for i in range(16):
            loss, gn = train_step(f, ema, opt, batch, largs, classes, i)
            hist.append((float(loss['overall'].detach()), float(gn)))
        return hist

    h_e, h_g = run(False), run(True)
    assert h_e == h_g, [(i, a, b) for i, (a, b) in enumerate(zip(h_e, h_g)) if a != b][:3]


@pytest.mark.gpu
def test_graphed_network_gradient_exchange_single_rank_nccl():
    """GraphedNetwork with the gradient exchange on, on a 1-rank RCCL group: parameter broadcast at construction, flat-bucket all-reduce
    (AVG over one rank = identity) on the static gradient buffers after every backward replay -- exactly the parameters of the plain eager
    run; a batch of another shape is refused (no silent un-exchanged fall-back); a DDP-wrapped module is refused."""
    import os
    import torch.distributed as dist
    from rsuper_amd.graph import GraphedNetwork
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import train_step, make_ema, wrap_ddp
    from rsuper_amd.training.utils import FusedAdamWEMA
    from rsuper_amd.hip import ops
    classes = synth.TINY_CLASSES
    bt = synth.batch(2, 32, classes, ['mask', 'report'], seed=5, diam_range=(4.0, 8.0), max_tumors=2)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
    batch['image'] = torch.from_numpy(synth.image(2, 32, seed=3)).to(DEV)
    la = argparse.Namespace(loss='ball_dice_both', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                            ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                            classification_branch=False, ema=True, ema_alpha=0.99)

    def run(graphed):
        torch.manual_seed(0)
        net = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='bf16').to(DEV)
        ema = make_ema(net)
        opt = FusedAdamWEMA(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        f = GraphedNetwork(net, warmup=2, exchange=True) if graphed else net
        for step in range(4):
            train_step(f, ema, opt, batch, la, classes, step)
        if graphed:
            assert f.exchange and f.fwd_graph is not None
            with pytest.raises(ValueError):
                f(batch['image'][:1])
        return [p.detach().clone() for p in net.parameters()]

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29543')
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        a = run(True)
        wrapped = wrap_ddp(UNet(1, 8, num_classes=len(classes), compute_dtype='bf16').to(DEV), 0)
        with pytest.raises(ValueError):
            GraphedNetwork(wrapped)
        red = getattr(wrapped, '_rsuper_reducer', None)
        if red is not None:
            red.remove()
    finally:
        dist.destroy_process_group()
        ops.GRAD_DEST = None
    b = run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['unet', 'medformer'])
def test_graphed_network_with_report_supervision_matches_eager(model):
    """rsuper_amd.graph.GraphedNetwork: forward and backward of the network replayed from two hipGraphs around an EAGER report-supervised
    loss (ball search with host reads) and the eager fused optimiser: exactly the losses, gradient norms, parameters and EMA of the eager
    train_step over different batches; evaluation calls and state_dict fall through to the module."""
    import argparse
    import synth
    from rsuper_amd.graph import GraphedNetwork
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    largs = argparse.Namespace(loss='ball_dice_both', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                               ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                               classification_branch=False, ema=True, ema_alpha=0.99)
    dev = 'cuda'
    batches = []
    for i in range(2):
        bt = synth.batch(2, 32, classes, ['mask', 'report'], seed=21 + i, diam_range=(4.0, 8.0), max_tumors=2)
        batches.append(dict(image=torch.from_numpy(synth.image(2, 32, seed=7 + i)).to(dev), label=torch.from_numpy(bt['label']).to(dev),
                            unk_channels=torch.from_numpy(bt['unk_channels']).to(dev), mask=torch.from_numpy(bt['mask']).to(dev),
                            volumes=torch.from_numpy(bt['volumes']).to(dev), diameters=torch.from_numpy(bt['diameters']).to(dev)))

    def run(graphed):
        torch.manual_seed(0)
        if model == 'unet':
            from rsuper_amd.model.dim3.unet import UNet
            net = UNet(1, 8, num_classes=len(classes), compute_dtype='bf16').to(dev)
        else:
            from rsuper_amd.model.dim3.medformer import MedFormer
            net = MedFormer(1, len(classes), compute_dtype='bf16', **{k: v for k, v in synth.MEDFORMER_TINY.items() if k not in ('size', 'seed')}).to(dev)
        ema = make_ema(net)
        opt = FusedAdamWEMA(net.parameters(), lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
        fwd = GraphedNetwork(net, warmup=2) if graphed else net
        hist = []
        for i in range(6):
            loss, gn = train_step(fwd, ema, opt, batches[i % 2], largs, classes, i)
            hist.append(({k: float(v.detach()) for k, v in loss.items()}, float(gn)))
        if graphed:
            assert fwd.fwd_graph is not None and set(fwd.state_dict().keys()) == set(net.state_dict().keys())
            # a second backward without zero_grad / optimiser step in between would overwrite the static gradient buffers: refused
            y2 = fwd(batches[0]['image'])['segmentation']
            with pytest.raises(RuntimeError):
                (y2[0] if isinstance(y2, (list, tuple)) else y2).sum().backward()
            opt.zero_grad(set_to_none=True)
            net.eval()
            with torch.no_grad():
                y = fwd(batches[0]['image'])['segmentation']          # evaluation falls through to the module
            net.train()
            assert torch.isfinite(y[0] if isinstance(y, (list, tuple)) else y).all()
        return hist, [p.detach().clone() for p in net.parameters()], [p.detach().clone() for p in ema.parameters()]

    h_e, p_e, e_e = run(False)
    h_g, p_g, e_g = run(True)
    assert h_e == h_g, (h_e, h_g)
    assert all(torch.equal(a, b) for a, b in zip(p_e, p_g)) and all(torch.equal(a, b) for a, b in zip(e_e, e_g))


@pytest.mark.gpu
@pytest.mark.parametrize('report', ['0', '0.1'])
def test_train_epoch_with_hip_graph_matches_eager(tmp_path, report):
    """train_net / train_epoch with --hip_graph: the epochs replayed from the captured step (segmentation-only supervision) or with the network
    forward / backward replayed from two graphs around the eager report-supervised loss give the same meter averages and the same final
    checkpointed weights as the eager driver."""
    import os
    from rsuper_amd.train_ddp import get_parser, main_worker
    from rsuper_amd.training.dataset import SyntheticUFODataset
    classes = ['kidney_left', 'kidney_right', 'liver', 'pancreas', 'pancreatic_lesion']
    out = {}
    for tag, extra in (('eager', []), ('graph', ['--hip_graph'])):
        ds = SyntheticUFODataset(classes, size=32, length=16, seed=3)
        args = get_parser(['--epochs', '2', '--batch_size', '2', '--cp_path', str(tmp_path) + '/', '--unique_name', tag, '--loss', 'ball_dice_last',
                           '--report_volume_loss_basic', report] + extra)
        args.base_chan, args.iter_per_epoch, args.print_freq, args.compute_dtype = 8, 4, 100, 'bf16'
        torch.manual_seed(0)
        hist = main_worker(0, 1, 0, args, trainset=ds)
        ck = torch.load(os.path.join(str(tmp_path), 'abdomenatlas_ufo', tag, 'fold_0_latest.pth'), map_location='cpu', weights_only=False)
        out[tag] = (hist, ck['model_state_dict'])
    drop = lambda h: [{k: v for k, v in e.items() if k != 'Elapsed Time'} for e in h]
    assert drop(out['eager'][0]) == drop(out['graph'][0]), (out['eager'][0], out['graph'][0])
    sd_e, sd_g = out['eager'][1], out['graph'][1]
    sd_e = sd_e.state_dict() if hasattr(sd_e, 'state_dict') else sd_e
    sd_g = sd_g.state_dict() if hasattr(sd_g, 'state_dict') else sd_g
    assert all(torch.equal(sd_e[k], sd_g[k]) for k in sd_e)


@pytest.mark.gpu
@pytest.mark.parametrize('report', ['0', '0.1'])
def test_train_epoch_with_hip_graph_on_a_packed_dataset_matches_eager(tmp_path, report):
    """--hip_graph on a dataset that yields bit-packed volumes (AugmentedCropDataset(packed=True), here the synthetic stand-in): with segmentation-only
    supervision train_epoch keeps the label a PackedBits, and the captured step must hold it in a static packed buffer (the capture used to call
    .to / .clone on it); with report supervision (round 6: all three volumes stay packed) the network is replayed from two graphs around the eager loss, which
    reads the packed volumes.  Same meters and final weights as the eager driver on the same packed dataset, and as the eager driver on the unpacked one."""
    import os
    from rsuper_amd.train_ddp import get_parser, main_worker
    from rsuper_amd.training.dataset import SyntheticUFODataset
    classes = ['kidney_left', 'kidney_right', 'liver', 'pancreas', 'pancreatic_lesion']
    out = {}
    for tag, extra, packed in (('plain', [], False), ('eager', [], True), ('graph', ['--hip_graph'], True)):
        ds = SyntheticUFODataset(classes, size=32, length=16, seed=3, packed=packed)
        args = get_parser(['--epochs', '2', '--batch_size', '2', '--cp_path', str(tmp_path) + '/', '--unique_name', tag, '--loss', 'ball_dice_last',
                           '--report_volume_loss_basic', report] + extra)
        args.base_chan, args.iter_per_epoch, args.print_freq, args.compute_dtype = 8, 4, 100, 'bf16'
        torch.manual_seed(0)
        hist = main_worker(0, 1, 0, args, trainset=ds)
        ck = torch.load(os.path.join(str(tmp_path), 'abdomenatlas_ufo', tag, 'fold_0_latest.pth'), map_location='cpu', weights_only=False)
        sd = ck['model_state_dict']
        out[tag] = (hist, sd.state_dict() if hasattr(sd, 'state_dict') else sd)
    drop = lambda h: [{k: v for k, v in e.items() if k != 'Elapsed Time'} for e in h]
    for tag in ('eager', 'graph'):
        assert drop(out['plain'][0]) == drop(out[tag][0]), (tag, out['plain'][0], out[tag][0])
        assert all(torch.equal(out['plain'][1][k], out[tag][1][k]) for k in out['plain'][1]), tag


@pytest.mark.gpu
def test_update_output_layer_loads_the_pretrained_checkpoint_before_the_surgery(tmp_path):
    """--update_output_layer (train_ddp.py:552-590, medformer.py:224-319): the old-class network is built, --pretrained is loaded INTO it, then the heads are
    rebuilt for the new class list -- rows of classes both lists name carry the checkpoint's weights, everything below the heads is the checkpoint (ADVICE r05:
    the surgery used to run on a freshly initialised head); without a checkpoint it refuses."""
    from rsuper_amd.train_ddp import get_parser, init_network
    from rsuper_amd.model.utils import get_model
    old = sorted(['kidney_left', 'kidney_right', 'liver', 'pancreas', 'pancreatic_lesion'])
    new = sorted(old + ['spleen', 'kidney_lesion'])
    base = ['--epochs', '1', '--batch_size', '2', '--cp_path', str(tmp_path) + '/', '--unique_name', 'onk', '--model', 'medformer']
    a0 = get_parser(base)
    torch.manual_seed(5)
    net_old = get_model(a0, pretrain=a0.pretrain, classes=old)
    path = str(tmp_path / 'old.pth')
    torch.save({'epoch': 1, 'model_state_dict': net_old.state_dict()}, path)
    a1 = get_parser(base + ['--update_output_layer', '--pretrained', path])
    torch.manual_seed(6)                                     # another seed: whatever agrees with net_old afterwards came from the file
    net, ema = init_network(a1, classes=new, old_classes=old)
    sd_old, sd = net_old.state_dict(), net.state_dict()
    for k, v in sd_old.items():
        if k.startswith('outc.') or k.startswith('aux_out.'):
            continue
        assert torch.equal(v, sd[k].cpu()), k
    for head in ('outc', 'aux_out'):
        if f'{head}.weight' not in sd:
            continue
        assert sd[f'{head}.weight'].shape[0] == len(new)
        for c in old:
            assert torch.equal(sd[f'{head}.weight'][new.index(c)].cpu(), sd_old[f'{head}.weight'][old.index(c)]), (head, c)
            assert torch.equal(sd[f'{head}.bias'][new.index(c)].cpu(), sd_old[f'{head}.bias'][old.index(c)]), (head, c)
    a2 = get_parser(base + ['--update_output_layer'])
    with pytest.raises(ValueError, match='pretrained'):
        init_network(a2, classes=new, old_classes=old)


@pytest.mark.gpu
def test_dispatcher_ops_forward_only_and_autograd_paths_agree():
    """rsuper::maxpool2 / rsuper::head_conv through the dispatcher: the AutogradCUDA kernel (forward + autograd node) and the CUDA kernel
    alone (inference_mode skips the autograd key) give identical results, and gradients flow through the registered op."""
    from rsuper_amd.hip import ops
    torch.manual_seed(0)
    x = torch.randn(2, 8, 8, 8, 16, device=DEV).to(torch.bfloat16).requires_grad_(True)
    y, mr = torch.ops.rsuper.maxpool2(x)
    assert y.grad_fn is not None and not mr.requires_grad
    with torch.inference_mode():
        y2, mr2 = torch.ops.rsuper.maxpool2(x.detach())
    assert torch.equal(y.detach(), y2) and torch.equal(mr, mr2)
    y.float().sum().backward()
    assert x.grad is not None and float(x.grad.float().abs().sum()) > 0
    w = torch.randn(5, 16, 1, 1, 1, device=DEV).requires_grad_(True)
    b = torch.randn(5, device=DEV).requires_grad_(True)
    lo = ops.HeadFn.apply(y.detach(), w, b)                                     # the modules' spelling of torch.ops.rsuper.head_conv
    with torch.inference_mode():
        lo2 = torch.ops.rsuper.head_conv(y.detach(), w.detach(), b.detach())
    assert torch.equal(lo.detach(), lo2)
    lo.sum().backward()
    assert w.grad is not None and b.grad is not None


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['unet', 'nopool', 'medformer'])
def test_no_kernel_reads_an_unwritten_buffer(which):
    """Three training steps with report supervision while every torch.empty is NaN-filled (torch.utils.deterministic.fill_uninitialized_memory;
    tools/uninit_hunt.py in its own process -- the switch is global): a kernel that reads an element nobody wrote turns a loss or a gradient
    into NaN.  UNet (pooled and strided down blocks) and MedFormer (shipped configuration, 128^3, 42 classes)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'uninit_hunt.py'), which, 'bf16'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith('step ')]
    assert r.returncode == 0 and len(lines) == 3, r.stdout[-2000:]
    for l in lines:
        assert 'finite True' in l and 'first NaN module: []' in l and 'non-finite gradients: []' in l, l


@pytest.mark.gpu
def test_supervision_prefetch_changes_nothing():
    """calculate_loss(pre=prepare_report_supervision(...)) == calculate_loss(...): losses and gradient bit for bit, single head and deep supervision;
    a `pre` made from other tensors is ignored."""
    from rsuper_amd.training import losses_foundation as lf
    classes = synth.PANTS_CLASSES
    S = 48
    bt = synth.batch(2, S, classes, ['mask', 'report'], seed=21, diam_range=(5.0, 14.0), max_tumors=3)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in bt.items()}
    lg = torch.from_numpy(synth.logits(2, len(classes), S, seed=22)).to(DEV)
    for loss_name, deep in (('ball_dice_both', False), ('ball_dice_last', True)):
        args = argparse.Namespace(loss=loss_name, aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.1, volume_loss_tolerance=0.2,
                                  ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                                  classification_branch=False, ema=True, ema_alpha=0.99)

        def run(mode):
            x = lg.clone().requires_grad_(True)
            out = [x, x * 0.5] if deep else x
            pre = None
            if mode == 'pre':
                pre = lf.prepare_report_supervision(t['label'], t['unk_channels'], t['mask'], t['volumes'], t['diameters'], classes, args)
                assert pre is not None
            elif mode == 'foreign':                    # prepared from clones: other tensors -> must be ignored, not trusted
                pre = lf.prepare_report_supervision(t['label'].clone(), t['unk_channels'].clone(), t['mask'].clone(), t['volumes'], t['diameters'],
                                                    classes, args)
            res = lf.calculate_loss({'segmentation': out}, t['label'], t['unk_channels'], args, None, t['mask'], t['volumes'], t['diameters'], classes,
                                    pre=pre)
            res['overall'].backward()
            return {k: float(v.detach()) for k, v in res.items()}, x.grad.clone()
        r0, g0 = run('plain')
        for mode in ('pre', 'foreign'):
            r1, g1 = run(mode)
            assert r0 == r1, (loss_name, mode, r0, r1)
            assert torch.equal(g0, g1), (loss_name, mode)
        assert 'ball_loss_bce' in r0 and r0['ball_loss_bce'] > 0
    a0 = argparse.Namespace(loss='ball_dice_both', report_volume_loss_basic=0.0)
    assert lf.prepare_report_supervision(t['label'], t['unk_channels'], t['mask'], t['volumes'], t['diameters'], classes, a0) is None


@pytest.mark.gpu
def test_pointwise_prepack_never_serves_a_dead_parameters_fragments():
    """A parameter that dies leaves its address (and a version counter of 0) to the next one: the prepacked fragments of the dead parameter
    must not be served to its heir.  (Round 3: `test_medformer_shipped_config_128_42_classes` failed once in a while after other MedFormer
    tests in the same process -- the first forward of a fresh model multiplied some activations with the previous test's weights.)"""
    import gc
    from rsuper_amd.hip import ops
    ops._PW_SEEN.clear(); ops._PW_PACKED.clear(); ops._PW_TABLES.clear()
    x = torch.randn(64, 32, device=DEV)

    def make(seed):
        g = torch.Generator(device=DEV).manual_seed(seed)
        return torch.nn.Parameter(torch.randn(48, 32, device=DEV, generator=g))
    wa = make(1)
    ops.pointwise_gemm(x, wa, None, 0, torch.float32)          # packs by itself, becomes a candidate
    ops.pointwise_prepack(torch.float32)                       # table + arena
    assert (wa.data_ptr(), 0, ops._DT[torch.float32]) in ops._PW_PACKED
    ya = ops.pointwise_gemm(x, wa, None, 0, torch.float32)     # served from the arena
    assert torch.allclose(ya, x @ wa.t(), atol=1e-4, rtol=1e-4)
    ptr = wa.data_ptr()
    del wa, ya
    gc.collect()
    wb = make(2)
    if wb.data_ptr() != ptr:
        pytest.skip('the allocator did not recycle the address')
    ops.pointwise_prepack(torch.float32)                       # what a module does at the top of its forward
    yb = ops.pointwise_gemm(x, wb, None, 0, torch.float32)
    assert torch.allclose(yb, x @ wb.t(), atol=1e-4, rtol=1e-4), 'fragments of the dead parameter were used'
    # and without the prepack call in between (a bare op call): the lookup itself must notice the heir
    ops.pointwise_prepack(torch.float32)
    ops.pointwise_prepack(torch.float32)
    assert torch.allclose(ops.pointwise_gemm(x, wb, None, 0, torch.float32), x @ wb.t(), atol=1e-4, rtol=1e-4)
    ptr = wb.data_ptr()
    del wb, yb
    gc.collect()
    wc = make(3)
    if wc.data_ptr() == ptr:
        assert torch.allclose(ops.pointwise_gemm(x, wc, None, 0, torch.float32), x @ wc.t(), atol=1e-4, rtol=1e-4), 'stale fragments at lookup'
    ops._PW_SEEN.clear(); ops._PW_PACKED.clear(); ops._PW_TABLES.clear()


@pytest.mark.gpu
def test_accumulators_need_no_memset():
    """The three entry points that used to zero their destination with hipMemsetAsync -- a node that, captured in a hipGraph, wrote 0xC0 bytes
    after eager interludes and turned the gradient norm into NaN -- now initialise it with kernels of their own: pre-filled with
    garbage (0xC0, as the defect did), the destinations must come out exactly as from a clean buffer."""
    import ctypes
    from rsuper_amd.hip import ops, lib as _l
    from rsuper_amd.training import losses_foundation as lf
    L = ops._L()
    st = torch.cuda.current_stream().cuda_stream
    # gradient norm accumulator (rsuper_grad_sqnorm): several chunks of tensors, accumulator holding 0xC0C0C0C0C0C0C0C0
    gs = [torch.randn(n, device=DEV) for n in (5, 4096 * 3 + 7, 1, 70000)] * 40            # 160 tensors: more than one launch chunk
    total = torch.empty(1, device=DEV, dtype=torch.float64)
    total.view(torch.uint8).fill_(0xC0)
    numel = (ctypes.c_size_t * len(gs))(*[g.numel() for g in gs])
    arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
    _l.check(L.rsuper_grad_sqnorm(len(gs), arr, numel, total.data_ptr(), st), 'grad_sqnorm')
    ref = sum(float((g.double() ** 2).sum()) for g in gs)
    assert abs(float(total) - ref) <= 1e-6 * ref, (float(total), ref)
    # plane_any flags
    m = torch.zeros((6, 4096), device=DEV, dtype=torch.uint8)
    m[1, 17] = 1; m[4, 4095] = 3
    flags = torch.full((6,), 0xC0, device=DEV, dtype=torch.uint8)
    _l.check(L.rsuper_plane_any(m.data_ptr(), 6, 4096, flags.data_ptr(), st), 'plane_any')
    assert flags.cpu().tolist() == [0, 1, 0, 0, 1, 0]
    # MaxPool3d(2) backward of an odd volume: the never-pooled trailing planes receive exactly zero
    x = torch.randn((1, 5, 7, 9, 8), device=DEV).to(torch.bfloat16)
    y, _ = ops._MaxPoolFn.apply(x.clone().requires_grad_(True)) if hasattr(ops, '_MaxPoolFn') else ops.MaxPoolFn.apply(x.clone().requires_grad_(True))
    dy = torch.ones_like(y)
    dx = torch.empty_like(x)
    dx.view(torch.uint8).fill_(0xC0)
    _l.check(L.rsuper_maxpool2_bwd(ops._DT[x.dtype], x.data_ptr(), 8, dy.data_ptr(), 8, dx.data_ptr(), 8, 1, 5, 7, 9, 8, st), 'maxpool2_bwd')
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx.float()).all()) and float(dx.float().sum()) == float(dy.float().sum())
    assert float(dx[:, 4].float().abs().sum()) == 0.0 and float(dx[:, :, 6].float().abs().sum()) == 0.0 and float(dx[:, :, :, 8].float().abs().sum()) == 0.0


def test_box_split_shape_respects_the_registered_workspace():
    """the split shape of the <= 6^3 box igemm writes nsplit x N x 216 x n_cols floats into the registered workspace; it may
    only be chosen while that fits (N x ceil(n_cols / 32) <= 256 with the default 7 MB), otherwise the launch takes another shape."""
    from rsuper_amd.hip import ops, lib
    L = ops._L()
    ws_bytes = L.rsuper_conv3_workspace_bytes()
    for N, cols in [(2, 640), (2, 320), (12, 640), (25, 320), (8, 1024)]:
        assert N * -(-cols // 32) <= 256
        assert L.rsuper_conv3_box_bn(lib.BF16, N, 6, 6, 6, cols) == 32, (N, cols)              # split shape (32-column blocks)
    for N, cols in [(13, 640), (26, 320), (32, 320), (32, 640), (64, 64), (300, 64)]:
        need1 = N * 216 * cols * 4                                                             # one split already exceeds the workspace?
        bn = L.rsuper_conv3_box_bn(lib.BF16, N, 6, 6, 6, cols)
        if need1 > ws_bytes:
            assert bn != 32, (N, cols, bn)
    # and the results of such launches are right (they would be even on an overflowing workspace, hence the shape assertions above)
    for a in [('bf16', 32, (6, 6, 6), 320, 0, 320, True), ('bf16', 13, (6, 6, 6), 64, 0, 320, True)]:
        r = gc.check_conv_bwd(*a)
        assert r['ok'], r


def test_step_guard_raises_one_step_late_and_nan_step_is_skipped():
    """The reference's per-step guards as device flags (train_ddp.StepGuard; train_ddp.py:311-313, losses_foundation.py:864-869, 1070-1071): a bad input
    or a NaN loss raises the reference's exception no later than one step after the fact, a clean step raises nothing, and the fused optimiser leaves
    parameters, moments and EMA untouched for a step whose gradient norm is not finite."""
    from rsuper_amd.train_ddp import StepGuard
    from rsuper_amd.training import losses_foundation as lf
    from rsuper_amd.training.utils import FusedAdamWEMA
    g = StepGuard(torch.device(DEV))
    img = torch.randn((2, 1, 16, 16, 16), device=DEV)
    g.check_input(img); g.end_step(); g.poll()                    # step 0 clean, nothing to report yet
    bad = img.clone(); bad[1, 0, 3, 4, 5] = 250.0
    g.check_input(bad); g.end_step(); g.poll()                    # step 1 is bad; poll() only looks at step 0
    g.check_input(img); g.end_step()
    with pytest.raises(AssertionError, match='Input is bigger than 100'):
        g.poll()                                                  # ... one step late
    g.poll(final=True)                                            # step 2 was clean
    nanimg = img.clone(); nanimg[0, 0, 0, 0, 1] = float('nan')
    g.check_input(nanimg); g.end_step()
    with pytest.raises(AssertionError, match='Input is nan'):
        g.poll(final=True)
    low = img.clone(); low[0, 0, 15, 15, 15] = -101.0
    g.check_input(low); g.end_step()
    with pytest.raises(AssertionError, match='smaller than -100'):
        g.poll(final=True)
    # calculate_loss routes its own guards through lf.GUARD instead of synchronising
    classes = synth.TINY_CLASSES
    bt = synth.batch(2, 32, classes, ['mask', 'report'], seed=5, diam_range=(4.0, 8.0), max_tumors=1)
    t = {k: torch.from_numpy(v).to(DEV) for k, v in bt.items()}
    x = torch.from_numpy(synth.logits(2, len(classes), 32, seed=6)).to(DEV)
    args = _args(loss='ball_dice_last', report_volume_loss_basic=0.1)
    old_s, old_g = lf.SANITY_CHECKS, lf.GUARD
    lf.SANITY_CHECKS, lf.GUARD = True, g
    try:
        xr = x.clone().requires_grad_(True)
        lf.calculate_loss({'segmentation': xr}, t['label'], t['unk_channels'], args, None, t['mask'], t['volumes'], t['diameters'], classes)
        g.end_step(); g.poll(final=True)                          # consistent batch: clean
        xn = x.clone(); xn[0, 0, 0, 0, 0] = float('nan')
        lf.calculate_loss({'segmentation': xn.requires_grad_(True)}, t['label'], t['unk_channels'], args, None, t['mask'], t['volumes'], t['diameters'], classes)
        g.end_step()
        with pytest.raises(ValueError, match='loss is nan'):
            g.poll(final=True)
        vol0 = torch.zeros_like(t['volumes'])
        lf.calculate_loss({'segmentation': x.clone().requires_grad_(True)}, t['label'], t['unk_channels'], args, None, t['mask'], vol0, torch.zeros_like(t['diameters']), classes)
        g.end_step()
        with pytest.raises(ValueError, match='tumor_volumes_report should not be all zeros'):
            g.poll(final=True)
    finally:
        lf.SANITY_CHECKS, lf.GUARD = old_s, old_g
    # optimiser: a non-finite gradient norm skips the update
    p = torch.nn.Parameter(torch.randn(1000, device=DEV))
    e = p.detach().clone()
    opt = FusedAdamWEMA([p], lr=1e-2)
    p.grad = torch.randn_like(p)
    opt.fused_step(max_norm=1.0, ema_params=[e], ema_alpha=0.5)
    ref_p, ref_e = p.detach().clone(), e.clone()
    st = opt._state_for(p)
    ref_m = {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
    p.grad = torch.full_like(p, float('nan'))
    opt.fused_step(max_norm=1.0, ema_params=[e], ema_alpha=0.5)
    torch.cuda.synchronize()
    assert torch.equal(p.detach(), ref_p) and torch.equal(e, ref_e)
    for k, v in ref_m.items():
        assert torch.equal(st[k], v), k
    p.grad = torch.randn_like(p)
    opt.fused_step(max_norm=1.0, ema_params=[e], ema_alpha=0.5)
    assert not torch.equal(p.detach(), ref_p) and bool(torch.isfinite(p).all())


@pytest.mark.gpu
def test_persistent_strided_forward_matches_parity_class_kernel():
    """csrc/conv3d_igemm_s2k.hip (the bf16 default of the strided [conv1 | shortcut] forward) against csrc/conv3d_igemm_s2.hip on the same operands --
    full size, odd sizes, ragged channel counts: outputs within bf16 roundings of each other (different fp32 summation orders), per-column
    statistics to 1e-3, and no further from the float64 evaluation of the same bf16 operands than the parity-class kernel is (tools/check_s2k.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_s2k.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), r.stdout[-2000:] + r.stderr[-2000:]
