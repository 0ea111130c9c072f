"""Host-side data formats of the training hot path: bit-packed label ingestion, epoch/rank sharding."""
from .packed import pack_bits, unpack_bits_device, ingest_packed_batch, PackedBits  # noqa: F401
from .sampler import ChunkedSampler  # noqa: F401
from .synthetic import SyntheticUFODataset  # noqa: F401
from .augmented import AugmentedCropDataset, save_crop, estimate_tumor_volume  # noqa: F401
