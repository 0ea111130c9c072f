"""3-D inference helpers mirroring rsuper_train/inference: forward-only reuse of the HIP conv stack."""
from .utils import get_inference, split_idx  # noqa: F401
from .inference3d import inference_whole_image, inference_sliding_window  # noqa: F401
