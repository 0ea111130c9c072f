"""inconv / down_block / up_block mirroring rsuper_train/model/dim3/unet_utils.py:7-75 (same module tree, so
state_dict keys are identical: inc.conv1.weight, down1.conv.1.conv1.conv.weight, up4.conv.0.shortcut.conv.weight...)."""
import torch
import torch.nn as nn

from .conv_layers import BasicBlock
from ...hip import ops


class _Pool(nn.Module):
    """Occupies index 0 of down_block.conv like nn.MaxPool3d does in the reference (unet_utils.py:36)."""

    def forward(self, x, with_skip=False):
        return ops.MaxPoolSkipFn.apply(x) if with_skip else ops.MaxPoolFn.apply(x)


class inconv(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), block=BasicBlock, norm='in'):
        super().__init__()
        self.conv1 = nn.Conv3d(in_ch, out_ch, kernel_size=3, padding=1, bias=False)   # holder (unet_utils.py:14)
        self.conv2 = block(out_ch, out_ch, kernel_size=kernel_size, norm=norm)

    def forward(self, img, dtype):
        x, mr = ops.StemFn.apply(img, self.conv1.weight, dtype)
        return self.conv2(x, mr)


class down_block(nn.Module):
    def __init__(self, in_ch, out_ch, num_block, block=BasicBlock, kernel_size=(3, 3, 3), down_scale=(2, 2, 2), pool=True, norm='in'):
        super().__init__()
        ds = tuple(down_scale) if isinstance(down_scale, (list, tuple)) else (down_scale,) * 3
        if ds != (2, 2, 2):
            raise NotImplementedError('gfx950 hot path implements down_scale (2, 2, 2) (every shipped config)')
        self.pool = bool(pool)
        if pool:                                               # unet_utils.py:35-37 (what get_model builds, model/utils.py:87-89)
            layers = [_Pool(), block(in_ch, out_ch, kernel_size=kernel_size, norm=norm)]
        else:                                                  # unet_utils.py:38-39: strided first block, no pooling layer
            layers = [block(in_ch, out_ch, kernel_size=kernel_size, stride=2, norm=norm)]
        for _ in range(num_block - 1):
            layers.append(block(out_ch, out_ch, kernel_size=kernel_size, norm=norm))
        self.conv = nn.Sequential(*layers)

    def forward(self, x, mr, with_skip=False):
        """with_skip: also return the input as the tensor to hand to the matching up_block (pooling layers only: its gradient is then added inside
        the max-pool backward kernel; with a strided first block the input itself is returned)."""
        blocks = list(self.conv)
        skip = x
        if self.pool:
            if with_skip and torch.is_grad_enabled() and x.requires_grad:
                x, mr, skip = blocks[0](x, True)
            else:
                x, mr = blocks[0](x)
            blocks = blocks[1:]
        for blk in blocks:
            x, mr = blk(x, mr)
        return (x, mr, skip) if with_skip else (x, mr)


class up_block(nn.Module):
    def __init__(self, in_ch, out_ch, num_block, block=BasicBlock, kernel_size=(3, 3, 3), up_scale=(2, 2, 2), norm='in'):
        super().__init__()
        layers = [block(in_ch + out_ch, out_ch, kernel_size=kernel_size, norm=norm)]
        for _ in range(num_block - 1):
            layers.append(block(out_ch, out_ch, kernel_size=kernel_size, norm=norm))
        self.conv = nn.Sequential(*layers)

    def forward(self, x1, mr1, x2, mr2):
        """x1: coarse map, x2: skip.  cat([x2, x1]) (unet_utils.py:71) is never materialised: the first block
        reads both sources."""
        up, mru = ops.UpsampleFn.apply(x1, tuple(x2.shape[1:4]))
        x, mr = self.conv[0](x2, mr2, up, mru)
        for blk in list(self.conv)[1:]:
            x, mr = blk(x, mr)
        return x, mr
