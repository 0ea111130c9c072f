"""3D UNet / ResUNet with the constructor of rsuper_train/model/dim3/unet.py:12-47, running on gfx950 kernels.

Differences from the reference class, both required by the drop-in boundary:
  * forward returns {'segmentation': logits} -- the contract calculate_loss reads
    (training/losses_foundation.py:859; the reference UNet returns a bare tensor and cannot be trained by
    train_ddp.py as shipped);
  * `compute_dtype`: 'bf16' (default; bf16 activations, f32 accumulate, f32 logits) or 'f32' (parity mode).
"""
import os

import torch
import torch.nn as nn

from .conv_layers import BasicBlock
from .unet_utils import inconv, down_block, up_block
from ...hip import ops, lib as _lib


def _get_block(name):
    if name != 'BasicBlock':
        raise NotImplementedError(f"block={name!r}: the gfx950 hot path implements 'BasicBlock' (config/abdomenatlas/resunet_3d.yaml:13)")
    return BasicBlock


class UNet(nn.Module):
    def __init__(self, in_ch, base_ch, scale=[2, 2, 2, 2], kernel_size=[3, 3, 3, 3, 3], num_classes=1, block='BasicBlock',
                 pool=True, norm='in', compute_dtype=None):
        super().__init__()
        if norm != 'in':
            raise NotImplementedError("norm must be 'in' (InstanceNorm3d) on the gfx950 hot path")
        if in_ch != 1 or base_ch % 8:
            raise NotImplementedError('in_ch must be 1 and base_ch a multiple of 8')
        if base_ch not in (8, 16, 32, 64) or num_classes > 256:
            raise NotImplementedError('the stem / head kernels are instantiated for base_ch 8 / 16 / 32 / 64 and hold the weights of up to 256 classes '
                                      'in LDS (reference configs: base 32, 26-42 classes)')
        blk = _get_block(block)
        num_block = 2
        ks = [k if isinstance(k, (list, tuple)) else [k] * 3 for k in kernel_size]
        b = base_ch
        self.inc = inconv(in_ch, b, block=blk, kernel_size=ks[0], norm=norm)
        self.down1 = down_block(b, 2 * b, num_block, blk, ks[1], scale[0], pool, norm)
        self.down2 = down_block(2 * b, 4 * b, num_block, blk, ks[2], scale[1], pool, norm)
        self.down3 = down_block(4 * b, 8 * b, num_block, blk, ks[3], scale[2], pool, norm)
        self.down4 = down_block(8 * b, 10 * b, num_block, blk, ks[4], scale[3], pool, norm)
        self.up1 = up_block(10 * b, 8 * b, num_block, blk, ks[3], scale[3], norm)
        self.up2 = up_block(8 * b, 4 * b, num_block, blk, ks[2], scale[2], norm)
        self.up3 = up_block(4 * b, 2 * b, num_block, blk, ks[1], scale[1], norm)
        self.up4 = up_block(2 * b, b, num_block, blk, ks[0], scale[0], norm)
        self.outc = nn.Conv3d(b, num_classes, kernel_size=1)          # holder: weight (K,C,1,1,1) + bias
        self.compute_dtype = compute_dtype or os.environ.get('RSUPER_DTYPE', 'bf16')
        self.pool = bool(pool)

    def _dtype(self):
        return {'bf16': torch.bfloat16, 'f32': torch.float32}[self.compute_dtype]

    def _blocks_with_geometry(self, shape):
        """(block, Ca, Cb, spatial size) for every BasicBlock in execution order."""
        N, _, D, H, W = shape
        out = []
        b = self.inc.conv2.conv1.conv.weight.shape[0]
        out.append((self.inc.conv2, b, 0, (D, H, W)))
        size = (D, H, W)
        sizes = [size]
        for dn in (self.down1, self.down2, self.down3, self.down4):
            size = tuple(s // 2 for s in size)
            sizes.append(size)
            blks = list(dn.conv)[1:]
            cin = blks[0].conv1.conv.weight.shape[1]
            for i, blk in enumerate(blks):
                out.append((blk, cin if i == 0 else blk.conv1.conv.weight.shape[1], 0, size))
        for lvl, up in zip((3, 2, 1, 0), (self.up1, self.up2, self.up3, self.up4)):
            blks = list(up.conv)
            cout = blks[0].conv1.conv.weight.shape[0]
            cin_total = blks[0].conv1.conv.weight.shape[1]
            for i, blk in enumerate(blks):
                out.append((blk, cout if i == 0 else blk.conv1.conv.weight.shape[1], cin_total - cout if i == 0 else 0, sizes[lvl]))
        return out

    def _pack_all(self, shape, dt):
        """Re-order every conv weight into MFMA fragment order with ONE launch (forward + backward buffers)."""
        N = shape[0]
        with_bwd = torch.is_grad_enabled()
        specs, meta = [], []
        for blk, Ca, Cb, (D, H, W) in self._blocks_with_geometry(shape):
            tiles_total = ops._L().rsuper_conv3_tiles(D, H, W) * N
            w1, w2, ws = blk.weights()
            sp, bns = ops.block_pack_specs(w1, w2, ws, Ca, Cb, dt, tiles_total, with_bwd, (N, D, H, W))
            meta.append((blk, len(specs), len(sp), bns))
            specs += sp
        with torch.no_grad():
            bufs = ops.pack_weights_batch(dt, specs)
        for blk, i0, k, bns in meta:
            blk._packs = (bufs[i0:i0 + k], bns)

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.RSuperHipError('rsuper_amd UNet runs on MI355X only (no CPU fallback); move the input to cuda')
        _lib.require_device()
        dt = self._dtype()
        # Batched packing (one launch for all layers) is opt-in: measured SLOWER (+1.2 ms/step) than packing each layer
        # right before its convolution, because fragments packed up front are cold in L2/MALL when finally used.
        batch_pack = os.environ.get('RSUPER_BATCH_PACK', '0') == '1' and self.pool
        if batch_pack:
            self._pack_all(tuple(x.shape), dt)
        x1, m1 = self.inc(x, dt)
        x2, m2, x1 = self.down1(x1, m1, True)      # x_k comes back as the skip tensor: both of its gradients meet in the pooling layer's backward
        x3, m3, x2 = self.down2(x2, m2, True)
        x4, m4, x3 = self.down3(x3, m3, True)
        x5, m5, x4 = self.down4(x4, m4, True)
        o, mo = self.up1(x5, m5, x4, m4)
        o, mo = self.up2(o, mo, x3, m3)
        o, mo = self.up3(o, mo, x2, m2)
        o, mo = self.up4(o, mo, x1, m1)
        logits = ops.HeadFn.apply(o, self.outc.weight, self.outc.bias)
        if batch_pack:
            for blk, *_ in self._blocks_with_geometry(tuple(x.shape)):
                blk._packs = None       # the buffers stay alive in each block's autograd context until backward
        return {'segmentation': logits}
