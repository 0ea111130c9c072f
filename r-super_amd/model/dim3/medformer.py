"""MedFormer with the constructor and state_dict of rsuper_train/model/dim3/medformer.py:81-175 -- the network R-Super actually
trains.  forward returns {'segmentation': logits} or {'segmentation': [logits, aux_logits]} with aux_loss
(prepare_return, medformer.py:205-222), which is what calculate_loss consumes.

Execution: conv stem, BasicBlock stages, the trilinear up-sampling in front of them and the 1x1x1 output head on the gfx950 kernels
(bf16 or f32 `compute_dtype`); patch merging, semantic maps and the bidirectional-attention stages as fp32 PyTorch-ROCm ops for
now (medformer_utils.py explains the split).  The classification / CLIP branches are baselines outside R-Super and are rejected.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import medformer_utils as _mu
from .medformer_utils import Feat, inconv, down_block, up_block, SemanticMapFusion, pointwise, upsample_trilinear
from ...hip import ops, lib as _lib

_DEFAULTS = dict(conv_num=[2, 1, 0, 0, 0, 1, 2, 2], trans_num=[0, 1, 2, 2, 2, 1, 0, 0], chan_num=[64, 128, 256, 320, 256, 128, 64, 32],
                 num_heads=[1, 4, 8, 16, 8, 4, 1, 1])


class MedFormer(nn.Module):
    def __init__(self, in_chan, num_classes, base_chan=32, map_size=[4, 8, 8], conv_block='BasicBlock', conv_num=None, trans_num=None,
                 chan_num=None, num_heads=None, fusion_depth=2, fusion_dim=320, fusion_heads=4, expansion=4, attn_drop=0., proj_drop=0.,
                 proj_type='depthwise', norm='in', act='relu', kernel_size=[3, 3, 3, 3], scale=[2, 2, 2, 2], aux_loss=False,
                 classification_branch=False, class_list_seg=None, class_list_cls=None, clip_branch=False, clip_feats=768,
                 compute_dtype=None):
        super().__init__()
        conv_num, trans_num = conv_num or _DEFAULTS['conv_num'], trans_num or _DEFAULTS['trans_num']
        chan_num, num_heads = chan_num or _DEFAULTS['chan_num'], num_heads or _DEFAULTS['num_heads']
        act_name = act if isinstance(act, str) else getattr(act, '__name__', str(act)).lower()
        if conv_block != 'BasicBlock' or norm != 'in' or act_name != 'relu' or proj_type != 'depthwise':
            raise NotImplementedError("MedFormer on gfx950: conv_block='BasicBlock', norm='in', act='relu', proj_type='depthwise' "
                                      '(config/abdomenatlas_ufo/medformer_3d.yaml)')
        if classification_branch or clip_branch:
            raise NotImplementedError('classification / CLIP branches are baselines outside the R-Super path (medformer.py:153-166)')
        if attn_drop or proj_drop:
            raise NotImplementedError('dropout is 0 in every shipped MedFormer configuration')
        flat = lambda v: [tuple(t) if isinstance(t, (list, tuple)) else (t,) * 3 for t in v]
        if any(k != (3, 3, 3) for k in flat(kernel_size)) or any(s != (2, 2, 2) for s in flat(scale)):
            raise NotImplementedError('kernel size 3 and scale 2 at every stage (shipped configuration)')
        if in_chan != 1 or base_chan % 8 or base_chan > 32 or num_classes > 64:
            raise NotImplementedError('in_chan 1, base_chan a multiple of 8 and <= 32, <= 64 classes (stem / head kernels)')
        c, h = chan_num, num_heads
        dh = [c[i] // h[i] for i in range(8)]
        self.inc = inconv(in_chan, base_chan)
        self.down1 = down_block(base_chan, c[0], conv_num[0], trans_num[0], map_generate=False)
        self.down2 = down_block(c[0], c[1], conv_num[1], trans_num[1], h[1], dh[1], expansion, map_size, True)
        self.down3 = down_block(c[1], c[2], conv_num[2], trans_num[2], h[2], dh[2], expansion, map_size, True)
        self.down4 = down_block(c[2], c[3], conv_num[3], trans_num[3], h[3], dh[3], expansion, map_size, True)
        self.map_fusion = SemanticMapFusion(c[1:4], fusion_dim, fusion_heads, depth=fusion_depth)
        self.up1 = up_block(c[3], c[4], conv_num[4], trans_num[4], h[4], dh[4], expansion, map_shortcut=True)
        self.up2 = up_block(c[4], c[5], conv_num[5], trans_num[5], h[5], dh[5], expansion, map_shortcut=True, no_map_out=True)
        self.up3 = up_block(c[5], c[6], conv_num[6], trans_num[6])
        self.up4 = up_block(c[6], c[7], conv_num[7], trans_num[7])
        self.aux_loss = bool(aux_loss)
        if aux_loss:
            self.aux_out = nn.Conv3d(c[5], num_classes, kernel_size=1)
        self.outc = nn.Conv3d(c[7], num_classes, kernel_size=1)
        self.compute_dtype = compute_dtype or os.environ.get('RSUPER_DTYPE', 'bf16')
        # RSUPER_MF_BLAS=cublas (opt-in): the fp32 library GEMMs that are left (81-token fusion transformer, aux head) through rocBLAS instead of
        # torch's default hipBLASLt -- 7.5 vs 18.5 us of host time per call.  It was the default in round 2 and is NOT any more: it is a
        # process-wide torch setting, and with the pointwise products on csrc/pointwise.hip it no longer pays.  (The replay corruption
        # once attributed to captured rocBLAS launches was a captured memset node: DESIGN.md 3.4c.)
        blas = os.environ.get('RSUPER_MF_BLAS', 'default')
        if blas != 'default' and torch.cuda.is_available() and hasattr(torch.backends.cuda, 'preferred_blas_library'):
            torch.backends.cuda.preferred_blas_library(blas)
            _mu.gemm_library.active = blas == 'cublas'      # long reductions switch back to hipBLASLt per call (medformer_utils.gemm_library)

    def _dtype(self):
        return {'bf16': torch.bfloat16, 'f32': torch.float32}[self.compute_dtype]

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.RSuperHipError('rsuper_amd MedFormer runs on MI355X only (no CPU fallback); move the input to cuda')
        _lib.require_device()
        dt = self._dtype()
        # operand dtype of the 1x1x1 GEMMs in the attention stages: fp32 (RSUPER_MF_GEMM_BF16=1 rounds the operands to bf16 in the bf16 mode --
        # measured SLOWER, 47-53 vs 44 ms/step: the casts around every GEMM cost more than the faster MFMA rate saves at these sizes)
        _mu.GEMM_DTYPE = dt if os.environ.get('RSUPER_MF_GEMM_BF16') == '1' else torch.float32
        _mu.GEMM_COMPUTE = dt                      # HIP pointwise GEMMs: bf16 MFMA operands in the bf16 mode, exact-f32 MFMA in the parity mode
        if _mu.HIP_POINTWISE and os.environ.get('RSUPER_MF_PREPACK', '1') == '1':
            ops.pointwise_prepack(dt)              # MFMA fragments of every 1x1x1 / linear weight of the attention stages: one launch per step
        x0 = self.inc(x, dt)
        x1, _ = self.down1(x0, dt)
        x2, m2 = self.down2(x1, dt)
        x3, m3 = self.down3(x2, dt)
        x4, m4 = self.down4(x3, dt)
        maps = self.map_fusion([m2, m3, m4])
        out, smap = self.up1(x4, x3, maps[2], maps[1], dt)
        out, smap = self.up2(out, x2, smap, maps[0], dt)
        aux = None
        if self.aux_loss:                                      # deep supervision head at 1/4 resolution, up-sampled (medformer.py:190-194)
            aux = upsample_trilinear(pointwise(out.g(), self.aux_out), x.shape[-3:], planar=True)
        out, smap = self.up3(out, x1, smap, None, dt)
        out, smap = self.up4(out, x0, smap, None, dt)
        feat, _ = out.cl(dt)
        logits = ops.HeadFn.apply(feat, self.outc.weight, self.outc.bias)
        return {'segmentation': [logits, aux] if self.aux_loss else logits}


def update_output_layer_onk(model, original_classes, new_classes, copy_pancreas=False):
    """Output-layer surgery for the public checkpoints (medformer.py:224-319, call site train_ddp.py:574-580 `--update_output_layer`): the 1x1x1
    heads `model.outc` and -- with deep supervision -- `model.aux_out` are rebuilt for `new_classes`; the row of every class that also exists in
    `original_classes` is copied from the old head, with `copy_pancreas` (the `--no_mask` runs) the remaining rows take the 'pancreatic_lesion' row,
    otherwise they keep the fresh nn.Conv3d initialisation.  A head that already has len(new_classes) outputs is left untouched, as in the reference.
    The new layers are created in the reference's order (outc, then aux_out) so a seeded run draws the same initial rows; they live on the device and
    in the dtype of the layers they replace.  The classification branch the reference also rewires is a baseline outside R-Super (not constructible
    here).  Returns the model."""
    def update_conv(old_conv, full_class_list):
        new_conv = nn.Conv3d(old_conv.in_channels, len(full_class_list), kernel_size=old_conv.kernel_size, stride=old_conv.stride, padding=old_conv.padding,
                             dilation=old_conv.dilation, groups=old_conv.groups, bias=old_conv.bias is not None)
        new_conv = new_conv.to(device=old_conv.weight.device, dtype=old_conv.weight.dtype)
        with torch.no_grad():
            for new_idx, new_cls in enumerate(full_class_list):
                src = None
                if new_cls in original_classes:
                    src = original_classes.index(new_cls)
                elif copy_pancreas:
                    src = original_classes.index('pancreatic_lesion')          # ValueError when the old list has no such class, as in the reference
                if src is not None:
                    new_conv.weight[new_idx] = old_conv.weight[src]
                    if old_conv.bias is not None:
                        new_conv.bias[new_idx] = old_conv.bias[src]
        return new_conv

    original_classes, new_classes = list(original_classes), list(new_classes)
    if model.outc.out_channels != len(new_classes):
        model.outc = update_conv(model.outc, new_classes)
    if hasattr(model, 'aux_out') and model.aux_loss and model.aux_out.out_channels != len(new_classes):
        model.aux_out = update_conv(model.aux_out, new_classes)
    if getattr(model, 'classification_branch', None) is not None and not isinstance(model.classification_branch, bool):
        raise NotImplementedError('classification branch heads are a baseline outside the R-Super path (medformer.py:297-317)')
    return model
