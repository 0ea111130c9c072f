"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's 3-D inference helpers, rsuper_train/inference/inference3d.py:
  inference_whole_image   :8-25    sigmoid(net(img))
  inference_sliding_window :28-107  windows of `window_size` at half-window stride (the last one clamped to the volume
                                    end, inference/utils.py:29-43), sigmoid probabilities summed on the CPU and divided by
                                    the per-voxel window count; volumes smaller than the window are zero-padded at the far
                                    end and cropped back; with a pancreas mask, windows without mask voxels contribute
                                    zeros (but still count).
Pinned by tests/golden/inference.npz (generated from the imported reference, tests/golden/gen_golden_inference.py).
`net` is any callable (B,1,d,h,w) f32 -> logits tensor / dict / tuple, e.g. oracle.unet_oracle.unet_forward.
"""
import numpy as np
import torch


def split_idx(half_win, size, i):
    """inference/utils.py:29-43."""
    start = half_win * i
    end = start + half_win * 2
    if end > size:
        start, end = size - half_win * 2, size
    return start, end


def _logits(pred):
    if isinstance(pred, dict):
        pred = pred['segmentation']
    while isinstance(pred, (tuple, list)):
        pred = pred[0]
    return pred


def inference_whole_image(net, img):
    with torch.no_grad():
        return torch.sigmoid(_logits(net(img)))


def inference_sliding_window(net, img, window_size, classes, pancreas=None):
    img = torch.as_tensor(img)
    if pancreas is not None:
        pancreas = torch.as_tensor(pancreas)
        while pancreas.dim() < img.dim():
            pancreas = pancreas.unsqueeze(0)
        assert pancreas.shape == img.shape
    B, C, D, H, W = img.shape
    wd, wh, ww = window_size
    origin = None
    if D < wd or H < wh or W < ww:
        origin = (D, H, W)
        img = torch.nn.functional.pad(img, (0, max(0, ww - W), 0, max(0, wh - H), 0, max(0, wd - D)))
        B, C, D, H, W = img.shape
    hd, hh, hw = wd // 2, wh // 2, ww // 2
    acc = torch.zeros((B, classes, D, H, W))
    cnt = torch.zeros((B, 1, D, H, W))
    with torch.no_grad():
        for i in range(D // hd):
            for j in range(H // hh):
                for k in range(W // hw):
                    d0, d1 = split_idx(hd, D, i)
                    h0, h1 = split_idx(hh, H, j)
                    w0, w1 = split_idx(hw, W, k)
                    if pancreas is None or float(pancreas[:, :, d0:d1, h0:h1, w0:w1].sum()) > 0:
                        pred = torch.sigmoid(_logits(net(img[:, :, d0:d1, h0:h1, w0:w1])))
                    else:
                        pred = torch.zeros((B, classes, wd, wh, ww))
                    acc[:, :, d0:d1, h0:h1, w0:w1] += pred
                    cnt[:, :, d0:d1, h0:h1, w0:w1] += 1
    acc /= cnt
    if origin is not None:
        acc = acc[:, :, :origin[0], :origin[1], :origin[2]]
    return acc


def window_counts(size, win):
    """Per-position window count along one axis (the 3-D count is the outer product of the three axes)."""
    half = win // 2
    c = np.zeros(size, np.int64)
    for i in range(size // half):
        s, e = split_idx(half, size, i)
        c[s:e] += 1
    return c
