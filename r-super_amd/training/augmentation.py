"""Online intensity augmentations of the pre-cropped training volumes -- the six transforms `load_augmented_data`
(rsuper_train/training/dataset/dim3/dataset_abdomenatlas_UFO.py:1047-1060) applies, each behind `np.random.random() < 0.3`.

Every function takes the (1, C, D, H, W) float32 volume the loader holds at that point and consumes the torch / numpy global
generators in the same order and with the same shapes as rsuper_train/training/augmentation.py, so a run seeded like the
reference draws the same parameters: with the same seeds the outputs agree to float32 rounding (tests/golden/loader.npz).

Differences in *how* (not what): the blur is applied as three 1-D passes (k taps each instead of k^3; zero padding makes the
separable form exact up to rounding -- see gaussian_blur), and gamma / contrast avoid the reference's (C, N) broadcast
temporaries for the single-channel volumes this path feeds them.
"""
import math

import torch
import torch.nn.functional as F


def _expect_volume(img):
    if img.dim() != 5 or img.shape[0] != 1:
        raise ValueError('expected a (1, C, D, H, W) volume, got %s' % (tuple(img.shape),))


def brightness_multiply(img, multiply_range=(0.7, 1.3)):
    """img * U(lo, hi), one factor per volume (augmentation.py:85-102, per_channel=False)."""
    _expect_volume(img)
    lo, hi = multiply_range
    assert hi > lo, 'Invalid range'
    factor = torch.rand(size=(1, 1, 1, 1, 1)) * (hi - lo) + lo
    return img * factor


def brightness_additive(img, std, mean=0.0):
    """img + N(mean, std), one offset per volume (augmentation.py:68-82, per_channel=False)."""
    _expect_volume(img)
    return img + torch.normal(mean, std, size=(1, 1, 1, 1, 1))


def gamma(img, gamma_range=(0.5, 2.0)):
    """Gamma curve on the volume normalised to [0, 1], then restored to its original mean / (unbiased) std
    (augmentation.py:105-137, per_channel=False, retain_stats=True).  The reference draws torch.rand(C, 1) exponents and
    lets them broadcast against the flattened (1, N) volume; for C == 1 -- the only case the loader produces -- that is a
    single exponent, which is what is implemented; C > 1 is rejected rather than silently reshaped."""
    _expect_volume(img)
    C = img.shape[1]
    if C != 1:
        raise ValueError('gamma: only single-channel volumes are on this path')
    flat = img.reshape(1, -1)
    lo, hi = flat.min(), flat.max()
    span = hi - lo
    mean, std = flat.mean(), flat.std()
    g = torch.rand(C, 1) * (gamma_range[1] - gamma_range[0]) + gamma_range[0]
    out = torch.pow((flat - lo) / span, g) * span + lo
    out = out - out.mean()
    out = out / out.std() * std + mean
    return out.reshape(img.shape)


def contrast(img, contrast_range=(0.65, 1.5)):
    """(img - mean) * U(lo, hi) + mean, clamped to the original [min, max] (augmentation.py:139-168, preserve_range=True)."""
    _expect_volume(img)
    C = img.shape[1]
    if C != 1:
        raise ValueError('contrast: only single-channel volumes are on this path')
    flat = img.reshape(1, -1)
    lo, hi = flat.min(), flat.max()
    mean = flat.mean()
    factor = torch.rand(C, 1) * (contrast_range[1] - contrast_range[0]) + contrast_range[0]
    out = torch.clamp((flat - mean) * factor + mean, min=lo, max=hi)
    return out.reshape(img.shape)


def gaussian_kernel_1d(kernel_size, sigma):
    """Normalised 1-D Gaussian taps on the integer grid -(k//2) .. k//2.  The reference builds the k^3 kernel
    exp(-(x^2+y^2+z^2)/(2 sigma^2)) / sum (augmentation.py:35-47); that is the outer product of three of these."""
    x = torch.arange(-(kernel_size // 2), kernel_size // 2 + 1, dtype=torch.float32)
    k = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def gaussian_blur(img, sigma_range=(0.5, 1.0)):
    """Zero-padded Gaussian blur, sigma ~ U(lo, hi), kernel size 2*ceil(3 sigma)+1 (augmentation.py:49-65).
    Applied separably: because the padding is zeros, convolving along W, H and D in turn equals the reference's single
    k^3 conv3d exactly in real arithmetic (float32 differences ~1e-7 relative)."""
    _expect_volume(img)
    if img.shape[1] != 1:
        raise ValueError('gaussian_blur: only single-channel volumes are on this path')
    sigma = torch.rand(1) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    ks = 2 * math.ceil(3 * sigma) + 1
    k = gaussian_kernel_1d(ks, sigma)
    p = ks // 2
    out = F.conv3d(img, k.view(1, 1, 1, 1, ks), padding=(0, 0, p))
    out = F.conv3d(out, k.view(1, 1, 1, ks, 1), padding=(0, p, 0))
    out = F.conv3d(out, k.view(1, 1, ks, 1, 1), padding=(p, 0, 0))
    return out


def gaussian_noise(img, std, mean=0.0):
    """img + N(0, 1) * std + mean, one draw per voxel (augmentation.py:16-18)."""
    return img + torch.randn(img.shape) * std + mean
