"""Modules mirroring rsuper_train/model/dim3/conv_layers.py (ConvNormAct :16-53, BasicBlock :71-94) so that
parameter names/shapes match the reference state_dict; the math runs in fused gfx950 kernels
(rsuper_amd.hip.ops.BasicBlockFn) instead of module-by-module ATen calls."""
import weakref

import torch
import torch.nn as nn

from ...hip import ops


class ConvNormAct(nn.Module):
    """Parameter holder for conv(act(norm(x))) with preact=True, norm=InstanceNorm3d(eps=1e-4, affine=False),
    act=ReLU, conv bias=False (conv_layers.py:22-43).  InstanceNorm has no state, so only `conv.weight` exists."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, norm='in', act='relu', preact=True):
        super().__init__()
        ks = tuple(kernel_size) if isinstance(kernel_size, (list, tuple)) else (kernel_size,) * 3
        st = tuple(stride) if isinstance(stride, (list, tuple)) else (stride,) * 3
        if ks != (3, 3, 3) or st not in ((1, 1, 1), (2, 2, 2)) or not preact or norm != 'in' or act != 'relu':
            raise NotImplementedError('gfx950 hot path implements 3x3x3 convolutions with stride 1 (shipped configuration) or 2 '
                                      '(pool=False down-sampling), pre-activation InstanceNorm+ReLU (conv_layers.py:46-51); '
                                      'LeakyReLU / other norms are not reachable from the reference UNet')
        self.conv = nn.Conv3d(in_ch, out_ch, kernel_size=3, stride=st[0], padding=1, bias=False)   # holder: weight + default init
        self.stride = st[0]
        self.preact = True


class BasicBlock(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=1, norm='in', act='relu', preact=True):
        super().__init__()
        st = tuple(stride) if isinstance(stride, (list, tuple)) else (stride,) * 3
        self.stride = st[0]
        self.conv1 = ConvNormAct(in_ch, out_ch, kernel_size, stride, 1, norm, act, preact)
        self.conv2 = ConvNormAct(out_ch, out_ch, kernel_size, 1, 1, norm, act, preact)
        self.shortcut = nn.Sequential()
        if st != (1, 1, 1) or in_ch != out_ch:
            self.shortcut = ConvNormAct(in_ch, out_ch, kernel_size, stride, 1, norm, act, preact)

    def weights(self):
        ws = self.shortcut.conv.weight if isinstance(self.shortcut, ConvNormAct) else None
        return self.conv1.conv.weight, self.conv2.conv.weight, ws

    def forward(self, xa, mra, xb=None, mrb=None):
        w1, w2, ws = self.weights()
        packs = getattr(self, '_packs', None)        # set by UNet.forward (whole-network batched weight packing)
        if packs is None and not torch.is_grad_enabled():
            packs = self._cached_forward_packs(xa, xb, w1, w2, ws)
        if self.stride == 2:
            assert xb is None
            return ops.BasicBlockFn.apply(xa, mra, None, None, w1, w2, ws, None, 2)
        return ops.BasicBlockFn.apply(xa, mra, xb, mrb, w1, w2, ws, packs, 1)

    def _cached_forward_packs(self, xa, xb, w1, w2, ws):
        """Inference (no_grad): the MFMA fragment buffers only change when the weights do, so sliding-window inference packs
        each layer once instead of once per window.  Keyed on the parameters' version counters, ops.WEIGHTS_EPOCH (the fused optimiser
        writes parameters through raw pointers) and the launch geometry."""
        N, D, H, W, Ca = xa.shape
        Cb = 0 if xb is None else xb.shape[-1]
        tiles_total = ops._L().rsuper_conv3_tiles(D, H, W) * N
        key = (ops.WEIGHTS_EPOCH, w1._version, w2._version, None if ws is None else ws._version, w1.data_ptr(), xa.dtype, Ca, Cb, tiles_total)
        hit = getattr(self, '_pack_cache', None)
        # ... and on the identity of the parameters: a replaced parameter can inherit address and version counter of the one it replaced
        if hit is not None and hit[0] == key and all(r() is w for r, w in zip(hit[2], (w1, w2, ws)) if w is not None):
            return hit[1]
        specs, bns = ops.block_pack_specs(w1, w2, ws, Ca, Cb, xa.dtype, tiles_total, False, (N, D, H, W))
        packs = (ops.pack_weights_batch(xa.dtype, specs), bns)
        self._pack_cache = (key, packs, tuple(weakref.ref(w) if w is not None else None for w in (w1, w2, ws)))
        return packs
