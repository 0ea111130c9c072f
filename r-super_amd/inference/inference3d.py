"""Mirror of rsuper_train/inference/inference3d.py on the MI355X path.

inference_whole_image (:8-25) and inference_sliding_window (:28-107) keep the reference's names, arguments and return
convention (probabilities after sigmoid, (B, classes, D, H, W)); the difference is where the work happens: the network is
the HIP UNet, the window probabilities are accumulated in HBM by `rsuper_window_accumulate` (the reference ships every
window to the host and adds there, :97-99) and the division by the window counter (:101) is `rsuper_window_normalize`
with the separable per-axis counts.  `to_cpu=True` (default) returns a CPU tensor like the reference; pass False to keep
the result on the device.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..hip import lib as _l
from .utils import split_idx


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _logits(pred):
    if isinstance(pred, dict):
        pred = pred['segmentation']
    if isinstance(pred, (tuple, list)):
        pred = pred[0]
    if isinstance(pred, (tuple, list)):
        pred = pred[0]
    return pred


def _device_of(net):
    p = next(net.parameters())
    if not p.is_cuda:
        raise _l.RSuperHipError('inference needs the network on an MI355X device (no CPU fallback)')
    return p.device


def inference_whole_image(net, img, args=None, to_cpu=False):
    """img: (B, C, D, H, W).  Returns sigmoid(net(img)) (:8-25); use when the whole image fits the window the network was
    trained with."""
    net.eval()
    dev = _device_of(net)
    with torch.no_grad():
        pred = _logits(net(img.to(dev)))
        B, K, D, H, W = pred.shape
        pred = pred.contiguous().float()
        out = torch.empty_like(pred)
        _l.check(_l.lib().rsuper_window_accumulate(pred.data_ptr(), out.data_ptr(), B * K, D, H, W, D, H, W, 0, 0, 0, 1, _stream()),
                 'window_accumulate')
    return out.cpu() if to_cpu else out


def _counts(size, win, dev):
    half = win // 2
    c = np.zeros(size, np.float32)
    for i in range(size // half):
        s, e = split_idx(half, size, i)
        c[s:e] += 1
    return torch.from_numpy(c).to(dev)


def inference_sliding_window(net, img, args, pancreas=None, to_cpu=True, window_batch=8):
    """img: (B, C, D, H, W); args.window_size = (d, h, w), args.classes = number of output channels.
    Windows overlap by half a window; pancreas: optional mask for pancreas-only inference (windows without mask voxels are
    skipped and contribute zeros, :83-93).  window_batch: windows per network call (an internal batching, results are
    identical).  Returns probabilities (B, classes, D, H, W)."""
    net.eval()
    dev = _device_of(net)
    if pancreas is not None:
        while len(pancreas.shape) < len(img.shape):
            pancreas = pancreas.unsqueeze(0)
        assert pancreas.shape == img.shape, f"Pancreas mask shape must match image shape, got {pancreas.shape} and {img.shape}"
    img = img.to(dev).float()
    B, C, D, H, W = img.shape
    win_d, win_h, win_w = args.window_size
    flag = False
    if D < win_d or H < win_h or W < win_w:
        flag = True
        diff_D, diff_H, diff_W = max(0, win_d - D), max(0, win_h - H), max(0, win_w - W)
        img = F.pad(img, (0, diff_W, 0, diff_H, 0, diff_D))
        if pancreas is not None:
            pancreas = F.pad(pancreas, (0, diff_W, 0, diff_H, 0, diff_D))
        origin_D, origin_H, origin_W = D, H, W
        B, C, D, H, W = img.shape
    half_win_d, half_win_h, half_win_w = win_d // 2, win_h // 2, win_w // 2
    K = args.classes
    L = _l.lib()
    pred_output = torch.zeros((B, K, D, H, W), device=dev, dtype=torch.float32)
    pan_cpu = None if pancreas is None else pancreas.detach().cpu()
    # The windows are independent (InstanceNorm is per sample), so `window_batch` of them go through the network together:
    # same numbers as one window at a time, but the low-resolution layers fill the chip (192^3 volume, 64 windows: 150 ms at 1,
    # 110 ms at 4, 105 ms at 8, 98 ms at 16).
    wb = max(1, int(window_batch))
    pending = []                                                       # (d0, h0, w0) of the windows waiting in the batch

    def flush():
        if not pending:
            return
        xs = torch.cat([img[:, :, d0:d0 + win_d, h0:h0 + win_h, w0:w0 + win_w] for d0, h0, w0 in pending], 0).contiguous()
        pred = _logits(net(xs)).contiguous().float()
        assert pred.shape == (B * len(pending), K, win_d, win_h, win_w), f'network output {tuple(pred.shape)} does not match the window / args.classes'
        for q, (d0, h0, w0) in enumerate(pending):
            pq = pred[q * B:(q + 1) * B]
            _l.check(L.rsuper_window_accumulate(pq.data_ptr(), pred_output.data_ptr(), B * K, win_d, win_h, win_w, D, H, W,
                                                d0, h0, w0, 0, _stream()), 'window_accumulate')
        pending.clear()

    with torch.no_grad():
        for i in range(D // half_win_d):
            for j in range(H // half_win_h):
                for k in range(W // half_win_w):
                    d0, d1 = split_idx(half_win_d, D, i)
                    h0, h1 = split_idx(half_win_h, H, j)
                    w0, w1 = split_idx(half_win_w, W, k)
                    if pan_cpu is not None and not bool(pan_cpu[:, :, d0:d1, h0:h1, w0:w1].sum() > 0):
                        continue                                   # skipped window: adds zeros, still counted below
                    pending.append((d0, h0, w0))
                    if len(pending) == wb:
                        flush()
        flush()
        cd, ch, cw = _counts(D, win_d, dev), _counts(H, win_h, dev), _counts(W, win_w, dev)
        _l.check(L.rsuper_window_normalize(pred_output.data_ptr(), cd.data_ptr(), ch.data_ptr(), cw.data_ptr(), B * K, D, H, W, _stream()),
                 'window_normalize')
    if flag:
        pred_output = pred_output[:, :, :origin_D, :origin_H, :origin_W]
    return pred_output.cpu() if to_cpu else pred_output
