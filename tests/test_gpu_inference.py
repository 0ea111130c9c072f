"""GPU parity of the sliding-window / whole-image inference mirror against the reference-generated golden
fixture (tests/golden/inference.npz) and the CPU oracle."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _net(dtype):
    from rsuper_amd.model.dim3.unet import UNet
    from oracle import unet_oracle as uo
    classes = synth.TINY_CLASSES
    net = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype=dtype)
    sd = synth.fill_state_dict(uo.unet_param_shapes(1, 8, len(classes)), 3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(DEV)


def _case(g, name):
    shape = tuple(int(v) for v in g[f'{name}_shape'])
    win = [int(v) for v in g[f'{name}_win']]
    box = g[f'{name}_box']
    pan = None
    if box[0][0] >= 0:
        pan = torch.zeros(shape)
        pan[box[0][0]:box[0][1], box[1][0]:box[1][1], box[2][0]:box[2][1]] = 1
    return torch.from_numpy(synth.volume(shape, int(g[f'{name}_seed'][0]))), win, pan


@pytest.mark.parametrize('name', ['a', 'pad', 'skip'])
def test_sliding_window_f32_matches_reference(name):
    """f32 HIP path == reference output (tolerance 1e-4 on probabilities, BASELINE north_star)."""
    from rsuper_amd.inference import inference_sliding_window
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'inference.npz'))
    img, win, pan = _case(g, name)
    args = argparse.Namespace(window_size=win, classes=len(synth.TINY_CLASSES))
    pred = inference_sliding_window(_net('f32'), img, args, pancreas=pan)
    assert not pred.is_cuda and tuple(pred.shape) == (1, len(synth.TINY_CLASSES)) + tuple(img.shape[2:])
    sub, step = synth.subsample(pred.numpy(), 8192)
    assert step == int(g[f'{name}_step'][0])
    np.testing.assert_allclose(sub, g[f'{name}_sub'], atol=1e-4)
    np.testing.assert_allclose(synth.summary(pred.numpy())[:2], g[f'{name}_summary'][:2], rtol=1e-4)


def test_sliding_window_bf16_close_and_on_device():
    from rsuper_amd.inference import inference_sliding_window
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'inference.npz'))
    img, win, _ = _case(g, 'a')
    args = argparse.Namespace(window_size=win, classes=len(synth.TINY_CLASSES))
    pred = inference_sliding_window(_net('bf16'), img, args, to_cpu=False)
    assert pred.is_cuda
    sub, _ = synth.subsample(pred.cpu().numpy(), 8192)
    # bf16 storage: probabilities of the random-init tiny UNet move by a few 1e-2; bounded and unbiased
    assert np.abs(sub - g['a_sub']).mean() < 3e-2 and abs(float(sub.mean()) - float(g['a_sub'].mean())) < 1e-2
    assert float(pred.min()) >= 0.0 and float(pred.max()) <= 1.0


def test_whole_image_matches_reference_and_errors():
    from rsuper_amd.inference import inference_whole_image, get_inference
    from rsuper_amd.hip.lib import RSuperHipError
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'inference.npz'))
    net = _net('f32')
    out = inference_whole_image(net, torch.from_numpy(synth.volume((32, 32, 32), 7)))
    sub, _ = synth.subsample(out.cpu().numpy(), 4096)
    np.testing.assert_allclose(sub, g['whole_sub'], atol=1e-4)
    assert get_inference(argparse.Namespace(dimension='3d', sliding_window=False)) is inference_whole_image
    with pytest.raises(ValueError):
        get_inference(argparse.Namespace(dimension='4d', sliding_window=True))
    with pytest.raises(RSuperHipError):
        inference_whole_image(net.cpu(), torch.zeros(1, 1, 32, 32, 32))


def test_window_accumulate_rejects_out_of_range_window():
    from rsuper_amd.hip import lib
    acc = torch.zeros((1, 2, 8, 8, 8), device=DEV)
    lg = torch.zeros((1, 2, 4, 4, 4), device=DEV)
    rc = lib.lib().rsuper_window_accumulate(lg.data_ptr(), acc.data_ptr(), 2, 4, 4, 4, 8, 8, 8, 6, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    assert rc != 0


def test_inference_pack_cache_follows_fused_optimizer_updates():
    """Cached inference fragments must not survive a fused optimiser step (it writes parameters through raw pointers, which
    tensor._version does not see): eval -> train step -> eval gives the new weights' output, identical to a fresh module."""
    from rsuper_amd.model.dim3.unet import UNet
    from rsuper_amd.train_ddp import train_step, make_ema
    from rsuper_amd.training.utils import FusedAdamWEMA
    classes = synth.TINY_CLASSES
    net = _net('f32')
    x = torch.from_numpy(synth.volume((32, 32, 32), 11)).to(DEV)
    with torch.no_grad():
        y0 = net(x)['segmentation'].clone()
        y0b = net(x)['segmentation'].clone()          # second call uses the cached fragments
    assert torch.equal(y0, y0b)
    B, S = 1, 32
    bt = synth.batch(B, S, classes, ['mask'], seed=5)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in bt.items() if k in ('label', 'unk_channels', 'mask', 'volumes', 'diameters')}
    batch['image'] = torch.from_numpy(synth.image(B, S, seed=3)).to(DEV)
    la = argparse.Namespace(loss='ball_dice_last', aux_weight=[0.5, 0.5], seg_loss=1.0, report_volume_loss_basic=0.0, volume_loss_tolerance=0.2,
                            ball_bce_weight=1.0, ball_dice_weight=1.0, ball_volume_margin=0.2, multi_ch_tumor=False, stardard_ce_ball=False,
                            classification_branch=False, ema=True, ema_alpha=0.99)
    opt = FusedAdamWEMA(net.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    train_step(net, make_ema(net), opt, batch, la, classes, 0)
    with torch.no_grad():
        y1 = net(x)['segmentation'].clone()
    fresh = UNet(1, 8, num_classes=len(classes), block='BasicBlock', norm='in', compute_dtype='f32').to(DEV)
    fresh.load_state_dict(net.state_dict())
    with torch.no_grad():
        y2 = fresh(x)['segmentation']
    assert not torch.equal(y1, y0) and torch.equal(y1, y2)


def test_sliding_window_with_medformer_matches_oracle():
    """The inference mirror drives any network with the {'segmentation': ...} contract: MedFormer with deep supervision (only the final head
    is used, inference3d.py:85-90).  Probabilities of the f32 HIP path against the window-by-window CPU evaluation of the MedFormer oracle."""
    from rsuper_amd.inference import inference_sliding_window
    from rsuper_amd.model.dim3.medformer import MedFormer
    from oracle import medformer_oracle as mo
    cfg = synth.MEDFORMER_TINY
    net = MedFormer(1, len(synth.TINY_CLASSES), compute_dtype='f32', **{k: v for k, v in cfg.items() if k not in ('size', 'seed')})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sdn = synth.fill_state_dict(shapes, cfg['seed'])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()})
    net = net.to(DEV).eval()
    img = torch.from_numpy(synth.volume((32, 48, 32), 21))
    args = argparse.Namespace(window_size=[32, 32, 32], classes=len(synth.TINY_CLASSES))
    pred = inference_sliding_window(net, img, args)
    assert tuple(pred.shape) == (1, len(synth.TINY_CLASSES), 32, 48, 32) and bool(torch.isfinite(pred).all())
    # expected: the inference oracle (window schedule pinned by tests/golden/inference.npz) driving the MedFormer oracle window by window
    from oracle import inference_oracle as io
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    with torch.no_grad():
        ref = io.inference_sliding_window(lambda x: mo.medformer_forward(sd, x, cfg), img, [32, 32, 32], len(synth.TINY_CLASSES))
    np.testing.assert_allclose(pred.numpy(), ref.numpy(), atol=1e-4)
