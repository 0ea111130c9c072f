"""ctypes binding of csrc/librsuper_hip.so (C ABI declared in include/rsuper_hip.h).

The product path has no CPU fallback: `lib()` raises if the shared library is missing and
`require_device()` raises if device 0 is not a gfx950 GPU.
"""
import ctypes
import os
from ctypes import c_int, c_long, c_longlong, c_size_t, c_float, c_double, c_void_p, c_uint32, c_uint

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(os.path.dirname(_HERE), 'csrc', 'librsuper_hip.so')

F32, BF16 = 0, 1
_LIB = None

P = c_void_p
_SIGS = {
    'rsuper_version': (ctypes.c_char_p, []),
    'rsuper_device_check': (c_int, []),
    'rsuper_conv3_packed_elems': (c_size_t, [c_int] * 5),
    'rsuper_conv3_pack_weights': (c_int, [c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    'rsuper_conv3_pack_weights_batch': (c_int, [c_int, c_int, P, P, P, P, P, P]),
    'rsuper_conv3_tiles': (c_int, [c_int] * 3),
    'rsuper_conv3_variant': (c_int, [c_int]),
    'rsuper_conv3_wgrad2_min_tiles': (c_int, [c_int]),
    'rsuper_conv3_box_bn': (c_int, [c_int] * 6),
    'rsuper_conv3_kd_bn': (c_int, [c_int] * 8),
    'rsuper_conv3_set_workspace': (c_int, [P, c_size_t]),
    'rsuper_conv3_workspace_bytes': (c_size_t, []),
    'rsuper_conv3_wgrad_splits': (c_int, [c_int] * 9),
    'rsuper_conv3_part_rows': (c_int, [c_int] * 9),
    'rsuper_conv3_s2_part_rows': (c_int, [c_int] * 9),
    'rsuper_conv3_igemm_s2': (c_int, [c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, P, P]),
    'rsuper_conv3_igemm': (c_int, [c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                   P, c_int, P, c_int, P, P, c_int, c_int, P, P, c_int, c_int, P, P]),
    'rsuper_conv3_igemm_split_out': (c_int, [c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                             P, c_int, P, c_int, P, P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_longlong, P]),
    'rsuper_conv3_wgrad': (c_int, [c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_int,
                                   P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_conv3_wgrad_partial': (c_int, [c_int, c_int, P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_int,
                                           P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_conv3_wgrad_reduce': (c_int, [P, c_int, c_int, c_int, c_int, P, P, P]),
    'rsuper_conv3_wgrad_reduce_batch': (c_int, [c_int, P, P, P, P, P, P, P, P]),
    'rsuper_conv3_wgrad_reduce_batch_stats': (c_int, [c_int, P, P, P, P, P, P, P, c_int, P, P, P, P, P, P, P, P, c_float, P]),
    'rsuper_conv3_wgrad_s2_splits': (c_int, [c_int] * 7),
    'rsuper_conv3_wgrad_s2': (c_int, [c_int, P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_pointwise_packed_bytes': (c_size_t, [c_int, c_int, c_int]),
    'rsuper_pointwise': (c_int, [c_int, c_int, P, c_int, P, P, P, c_int, P, c_int, c_long, c_int, c_int, P, P]),
    'rsuper_pointwise_pack_batch': (c_int, [c_int, P, c_int, c_long, P, P]),
    'rsuper_pointwise_wgrad_splits': (c_int, [c_long, c_int, c_int]),
    'rsuper_pointwise_wgrad': (c_int, [c_int, P, c_int, P, c_int, c_long, c_int, c_int, P, c_int, P, P, P]),
    'rsuper_stats_finalize': (c_int, [P, c_int, c_int, c_int, c_double, c_float, c_int, c_int, P, P]),
    'rsuper_in_bwd_finalize': (c_int, [c_int, P, c_int, P, c_int, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    'rsuper_maxpool2_fwd': (c_int, [c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_maxpool2_bwd': (c_int, [c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_maxpool2_bwd_add': (c_int, [c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_subsample2_fwd': (c_int, [c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_subsample2_bwd': (c_int, [c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_upsample_fwd': (c_int, [c_int, P, c_int, P, c_int, P, c_int] + [c_int] * 8 + [P]),
    'rsuper_upsample_bwd': (c_int, [c_int, P, c_int, P, c_int] + [c_int] * 8 + [P]),
    'rsuper_stem_fwd': (c_int, [c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_stem_wgrad': (c_int, [c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_head_fwd': (c_int, [c_int, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    'rsuper_head_bwd_data': (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_head_bwd_weight': (c_int, [c_int, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    'rsuper_head_bwd': (c_int, [c_int, P, c_int, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    'rsuper_plane_partials_fwd': (c_int, [P, c_size_t, P, P, P, P, P, c_int, c_int, c_size_t, P]),
    'rsuper_plane_partials_fwd2': (c_int, [P, c_size_t, P, P, c_int, c_int, P, P, P, P, P, c_int, c_int, c_size_t, P]),
    'rsuper_plane_partials_fwd3': (c_int, [P, c_size_t, P, P, c_int, c_int, P, P, P, P, P, c_int, c_int, c_size_t, P]),
    'rsuper_plane_partials_blocks': (c_int, [c_size_t]),
    'rsuper_plane_sums_reduce': (c_int, [P, c_int, c_int, P, P]),
    'rsuper_plane_partials_bwd2': (c_int, [P, c_size_t, P, P, c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_size_t, P]),
    'rsuper_cnorm_rows': (c_int, [c_long]),
    'rsuper_cnorm_small': (c_int, [P, P, P, P, P, c_int, c_long, c_int, c_int, c_float, c_int, P]),
    'rsuper_cnorm_forward': (c_int, [P, P, P, P, c_int, c_long, c_int, c_int, c_float, P]),
    'rsuper_cnorm_backward': (c_int, [P, P, P, P, P, P, c_int, c_long, c_int, c_int, P]),
    'rsuper_cnorm_stats': (c_int, [P, P, P, P, c_int, c_long, c_int, c_int, c_int, P]),
    'rsuper_cnorm_apply': (c_int, [P, P, P, P, P, c_int, c_long, c_int, c_int, c_int, P]),
    'rsuper_token_attn_supported': (c_int, [c_int, c_int]),
    'rsuper_token_attn_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'rsuper_token_attn_bwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'rsuper_battn_supported': (c_int, [c_int, c_int, c_int]),
    'rsuper_battn_chunks': (c_int, [c_int, c_int]),
    'rsuper_battn_fwd': (c_int, [P] * 7 + [c_int] * 5 + [c_float, P]),
    'rsuper_battn_bwd': (c_int, [P] * 9 + [c_int] * 5 + [c_float, P]),
    'rsuper_se_forward': (c_int, [P] * 10 + [c_int, c_long, c_int, c_int, P]),
    'rsuper_se_backward': (c_int, [P] * 17 + [c_int, c_long, c_int, c_int, P]),
    'rsuper_cl_planar': (c_int, [P, P, c_int, c_long, c_int, c_int, c_int, P]),
    'rsuper_depthwise3_rows': (c_int, [c_long]),
    'rsuper_depthwise3_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_depthwise3_wgrad': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'rsuper_seg_from_sums': (c_int, [P, P, c_int, c_int, c_size_t, c_double, P, P, P]),
    'rsuper_report_from_sums': (c_int, [P, P, c_int, c_int, c_int, c_size_t, c_int, P, P, c_double, c_double, c_int, P, c_int, c_int, P, P, P]),
    'rsuper_plane_partials_bwd': (c_int, [P, c_size_t, P, P, P, P, P, P, c_int, c_int, c_size_t, P]),
    'rsuper_sigmoid_mask': (c_int, [P, P, P, c_size_t, P]),
    'rsuper_window_accumulate': (c_int, [P, P] + [c_int] * 11 + [P]),
    'rsuper_window_normalize': (c_int, [P, P, P, P, c_long, c_int, c_int, c_int, P]),
    'rsuper_dilate_volume': (c_int, [P, P, P, c_long, c_int, c_int, c_int, c_int, P]),
    'rsuper_dilate_volume_sparse': (c_int, [P, P, P, P, c_long, c_int, c_int, c_int, c_int, P]),
    'rsuper_ball_workspace_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'rsuper_ball_conv_argmax': (c_int, [P, c_int, c_int, c_int, c_int, c_float, P, P, P, P]),
    'rsuper_insert_ball': (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    'rsuper_insert_ball_at': (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P, P]),
    'rsuper_radix_hist': (c_int, [P, P, c_long, c_uint32, c_int, P, P]),
    'rsuper_topk_mark': (c_int, [P, P, c_long, c_uint32, c_uint, P, P]),
    'rsuper_compact': (c_int, [P, P, c_long, P, P, P, P]),
    'rsuper_rank_weights': (c_int, [P, P, c_uint, c_float, c_float, P, P]),
    'rsuper_rank_assign': (c_int, [P, c_uint, c_float, c_float, P, P]),
    'rsuper_topk_select': (c_int, [P, P, c_long, c_uint, P, P, P]),
    'rsuper_topk_select_multi': (c_int, [P, P, c_long, P, c_int, P, P, c_int, P]),
    'rsuper_plane_any': (c_int, [P, c_long, c_long, P, P]),
    'rsuper_guard_range': (c_int, [P, c_size_t, c_float, c_float, P, P]),
    'rsuper_guard_consistency': (c_int, [P, P, P, c_int, c_int, P, P]),
    'rsuper_mask_op': (c_int, [P, P, c_long, c_int, P]),
    'rsuper_unpack_bits': (c_int, [P, P, c_int, c_int, c_int, c_long, P]),
    'rsuper_timer_event_create': (c_int, [P]),
    'rsuper_timer_event_record': (c_int, [P, P]),
    'rsuper_timer_event_elapsed_ms': (c_int, [P, P, P]),
    'rsuper_timer_event_destroy': (c_int, [P]),
    'rsuper_unpack_bits_sel': (c_int, [P, P, c_int, c_int, c_int, c_long, P, P, P]),
    'rsuper_plane_any_bits': (c_int, [P, c_int, c_int, c_int, c_long, P, P]),
    'rsuper_zero_where': (c_int, [P, P, c_long, P]),
    'rsuper_count': (c_int, [P, c_long, P, P]),
    'rsuper_grad_sqnorm': (c_int, [c_int, P, P, P, P]),
    'rsuper_clip_scale': (c_int, [c_int, P, P, c_float, P, P]),
    'rsuper_adamw_ema_step': (c_int, [c_int, P, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_float, P, P]),
    'rsuper_adamw_ema_step_dyn': (c_int, [c_int, P, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, P, P, P]),
}

ERR = {1: 'RSUPER_ERR_ARG', 2: 'RSUPER_ERR_LAUNCH', 3: 'RSUPER_ERR_UNSUPPORTED', 4: 'RSUPER_ERR_NO_DEVICE'}


class RSuperHipError(RuntimeError):
    pass


def lib():
    """Load librsuper_hip.so; no fallback -- a missing library is a hard error."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RSuperHipError(f'{SO_PATH} not found: build it with `make -C r-super_amd/csrc` '
                                 f'(or python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.')
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)      # AttributeError here == header/library mismatch
            fn.restype, fn.argtypes = res, args
        if 'RSUPER_IGEMM_VARIANT' in os.environ:      # 0 = classic kernel, 1 = producer/consumer persistent kernel, 2 = per-launch choice (default)
            L.rsuper_conv3_variant(int(os.environ['RSUPER_IGEMM_VARIANT']))
        _LIB = L
    return _LIB


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what):
    if rc != 0:
        raise RSuperHipError(f'{what} failed: {ERR.get(rc, rc)}')


def require_device():
    check(lib().rsuper_device_check(), 'rsuper_device_check (MI355X / gfx950 required)')
