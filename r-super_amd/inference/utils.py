"""rsuper_train/inference/utils.py: get_inference (:4-25), split_idx (:29-43)."""


def get_inference(args):
    if args.dimension == '3d':
        if args.sliding_window:
            from .inference3d import inference_sliding_window
            return inference_sliding_window
        from .inference3d import inference_whole_image
        return inference_whole_image
    if args.dimension == '2d':
        raise NotImplementedError('2d inference is outside the accelerated hot path (3-D UNet only)')
    raise ValueError('Error in image dimension')


def split_idx(half_win, size, i):
    """half_win: size of half a window; size: image size along the axis; i: patch index.  The last window is clamped to
    the end of the volume (:39-41)."""
    start_idx = half_win * i
    end_idx = start_idx + half_win * 2
    if end_idx > size:
        start_idx = size - half_win * 2
        end_idx = size
    return start_idx, end_idx
