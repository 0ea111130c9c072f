"""Bit-packed label ingestion.

The reference dataset stores the label, unknown-channel and chosen-segment volumes of a crop as
`np.packbits(bool (C, D, H, W), axis=0)` (training/dataset/dim3/dataset_abdomenatlas_UFO.py:955,970,975), inflates them on
the host with `np.unpackbits(...)[:num_classes]` (:1031-1034) and the training loop then casts them to int64 / float32
before the H2D copy (train_ddp.py:249-261): 26 x 96^3 x (8 + 4 + 4) bytes per sample.  Here the packed bytes travel
(ceil(C/8) bytes per voxel and volume: 16x..64x less PCIe traffic) and are inflated on the device by the HIP kernel
`rsuper_unpack_bits` into the uint8 0/1 masks `calculate_loss` consumes directly.
"""
import numpy as np
import torch

from ...hip import lib as _l


def pack_bits(volume):
    """np.packbits(bool (..., C, D, H, W), axis=-4) as the dataset writer does (:955); accepts numpy or torch (CPU)."""
    a = volume.cpu().numpy() if isinstance(volume, torch.Tensor) else np.asarray(volume)
    return np.packbits(a.astype(np.bool_), axis=-4)


def unpack_bits_device(packed, num_classes, stream=None):
    """packed: uint8 CUDA tensor (B, P, D, H, W) or (P, D, H, W), P = ceil(C/8)  ->  uint8 0/1 tensor (B, C, D, H, W).
    Same result as np.unpackbits(packed, axis=channel)[:num_classes] (:1031-1034); asserts the reference's shape checks."""
    if packed.dim() == 4:
        packed = packed.unsqueeze(0)
    assert packed.dim() == 5 and packed.dtype == torch.uint8, 'packed volume must be uint8 (B, P, D, H, W)'
    if not packed.is_cuda:
        raise _l.RSuperHipError('unpack_bits_device needs a device tensor (no CPU fallback)')
    B, P = packed.shape[:2]
    assert P * 8 >= num_classes and P * 8 < num_classes + 10, 'packed channel count does not match num_classes (:1032-1033)'
    packed = packed.contiguous()
    V = packed[0, 0].numel()
    out = torch.empty((B, num_classes) + tuple(packed.shape[2:]), device=packed.device, dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    _l.check(_l.lib().rsuper_unpack_bits(packed.data_ptr(), out.data_ptr(), B, P, num_classes, V, st), 'unpack_bits')
    return out


_FORCE = {}      # (C, channels, device) -> (C,) uint8 device table of the channels PackedBits.planes always inflates


class PackedBits:
    """A label volume kept in the dataset's bit-packed form on the device: `packed` (B, P, D, H, W) uint8 as `np.packbits(axis = class)` wrote it, `C` classes.
    `calculate_loss` reads it directly in the segmentation term (csrc/loss.hip: the loss kernels extract bit 7 - (c & 7) of byte plane c >> 3; SURVEY 8f-2
    "loss kernels should read the packed u8 directly"); everything else gets the uint8 0/1 form from `unpack()` (cached)."""

    def __init__(self, packed, num_classes):
        assert packed.dim() == 5 and packed.dtype == torch.uint8 and packed.is_cuda
        P = packed.shape[1]
        assert P * 8 >= num_classes and P * 8 < num_classes + 10, 'packed channel count does not match num_classes (:1032-1033)'
        self.packed, self.C = packed.contiguous(), int(num_classes)
        self._u8, self._flags, self._planes = None, None, {}

    @property
    def shape(self):
        return torch.Size((self.packed.shape[0], self.C) + tuple(self.packed.shape[2:]))

    @property
    def device(self):
        return self.packed.device

    @property
    def is_cuda(self):
        return self.packed.is_cuda

    def unpack(self):
        if self._u8 is None:
            self._u8 = unpack_bits_device(self.packed, self.C)
        return self._u8

    # ---- what the report losses read of a volume without inflating it (csrc/morph.hip unpack_bits_sel / plane_any_bits) --------------------------------
    def _force(self, chs):
        key = tuple(int(c) for c in chs)
        f = _FORCE.get((self.C, key, self.packed.device))
        if f is None:
            h = torch.zeros(self.C, dtype=torch.uint8)
            h[list(key)] = 1
            f = _FORCE[(self.C, key, self.packed.device)] = h.to(self.packed.device)
        return f

    def class_flags(self):
        """(B * C,) uint8 device flags: class c of sample b has a voxel (np.unpackbits(...)[c].any() from the packed bytes, 1/8 of the traffic)."""
        if self._flags is None:
            B, P = self.packed.shape[:2]
            V = self.packed[0, 0].numel()
            if V % 16 == 0 and self.packed.data_ptr() % 16 == 0:
                self._flags = torch.empty(B * self.C, device=self.packed.device, dtype=torch.uint8)
                _l.check(_l.lib().rsuper_plane_any_bits(self.packed.data_ptr(), B, P, self.C, V, self._flags.data_ptr(), torch.cuda.current_stream().cuda_stream),
                         'plane_any_bits')
            else:       # voxel counts that are not a multiple of 16 (the kernel reads 16-byte vectors): per bit, the maximum of (byte & bit) over the plane
                bits = torch.stack([(self.packed.flatten(2) & (0x80 >> k)).amax(2) for k in range(8)], 2)   # (B, P, 8)
                self._flags = (bits.reshape(B, P * 8)[:, :self.C] != 0).to(torch.uint8).reshape(-1).contiguous()
        return self._flags

    def planes(self, chs, with_flagged=False):
        """uint8 (B, C, D, H, W) in which ONLY the planes of the channels `chs` (and, with_flagged, every plane that holds a voxel at all: class_flags) are
        written -- the rest of the tensor is never read by its users (the report losses index the lesion channels; the dilated unknown map is consumed
        through its flags).  Cached per (chs, with_flagged)."""
        key = (tuple(int(c) for c in chs), bool(with_flagged))
        t = self._planes.get(key)
        if t is None:
            B, P = self.packed.shape[:2]
            V = self.packed[0, 0].numel()
            t = torch.empty(tuple(self.shape), device=self.packed.device, dtype=torch.uint8)
            fl = self.class_flags() if with_flagged else None
            _l.check(_l.lib().rsuper_unpack_bits_sel(self.packed.data_ptr(), t.data_ptr(), B, P, self.C, V, fl.data_ptr() if fl is not None else None,
                                                     self._force(chs).data_ptr(), torch.cuda.current_stream().cuda_stream), 'unpack_bits_sel')
            self._planes[key] = t
        return t

    def sample_any(self, as_bool=True):
        """(B,) any voxel of any class per sample (the `.any()` tests of calculate_loss :864-869) from the packed bytes."""
        B = self.packed.shape[0]
        n = self.packed[0].numel()
        f = torch.empty(B, device=self.packed.device, dtype=torch.uint8)
        if n % 16 == 0 and self.packed.data_ptr() % 16 == 0:
            _l.check(_l.lib().rsuper_plane_any(self.packed.data_ptr(), B, n, f.data_ptr(), torch.cuda.current_stream().cuda_stream), 'plane_any')
        else:
            f = self.packed.flatten(1).any(1).to(torch.uint8)
        return f.bool() if as_bool else f

    def reset(self):
        """Forget everything derived from `packed` (a hipGraph step copies the next batch into the same buffer)."""
        self._u8, self._flags = None, None
        self._planes.clear()


def ingest_packed_batch(sample, num_classes, device='cuda', keep_label_packed=False, keep_packed=False):
    """sample: dict with 'image' (B,1,D,H,W) f32 and bit-packed uint8 'label', 'unk_channels', 'mask' (B,P,D,H,W), plus
    'volumes', 'diameters' [, 'weights'] as the dataset yields them.  Returns the device batch `train_step` takes, with the
    three volumes as uint8 0/1 (B,C,D,H,W).  keep_label_packed: 'label' stays a `PackedBits` -- the loss kernels read the
    bits (1/8 of the label bytes cross HBM, no inflated copy is written).  keep_packed: all three volumes stay `PackedBits` (round 6): `calculate_loss`
    reads the label bits in the segmentation term, inflates only the lesion planes the report losses index (`PackedBits.planes`) and takes the unknown
    map's plane flags from the packed bytes -- with or without report supervision."""
    out = {}
    for k, v in sample.items():
        t = torch.as_tensor(v)
        if k in ('label', 'unk_channels', 'mask') and (keep_packed or (k == 'label' and keep_label_packed)):
            p = t.to(device, non_blocking=True)
            out[k] = PackedBits(p.unsqueeze(0) if p.dim() == 4 else p, num_classes)
        elif k in ('label', 'unk_channels', 'mask'):
            out[k] = unpack_bits_device(t.to(device, non_blocking=True), num_classes)
        else:
            out[k] = t.to(device, non_blocking=True)
            if k in ('volumes', 'diameters', 'weights', 'image'):
                out[k] = out[k].float()
    return out
