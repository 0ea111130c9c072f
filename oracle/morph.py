"""ctypes wrapper over oracle/morph.c (ORACLE -- test infrastructure only)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, '_build', 'liboracle_morph.so')
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, 'morph.c')):
            subprocess.check_call(['make', '-C', _HERE, '-s'])
        L = ctypes.CDLL(so)
        L.oracle_ball_nnz.restype = ctypes.c_int
        L.oracle_ball_nnz.argtypes = [ctypes.c_int]
        L.oracle_dilate_volume.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 4
        L.oracle_topk_mask.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
        L.oracle_rank_desc.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        _LIB = L
    return _LIB


def ball_nnz(k):
    return lib().oracle_ball_nnz(int(k))


def dilate_volume(vol, kernel_size):
    """vol: uint8/bool ndarray (..., D, H, W) of 0/1.  losses_foundation.py:22-99."""
    v = np.ascontiguousarray(vol, dtype=np.uint8)
    D, H, W = v.shape[-3:]
    nvol = int(np.prod(v.shape[:-3])) if v.ndim > 3 else 1
    out = np.empty_like(v)
    lib().oracle_dilate_volume(v.ctypes.data, out.ctypes.data, nvol, D, H, W, int(kernel_size))
    return out


def topk_mask(x, k):
    x = np.ascontiguousarray(x, dtype=np.float32)
    m = np.empty(x.size, np.uint8)
    lib().oracle_topk_mask(x.ctypes.data, x.size, int(k), m.ctypes.data)
    return m.reshape(x.shape)


def rank_desc(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    r = np.empty(x.size, np.int64)
    lib().oracle_rank_desc(x.ctypes.data, x.size, r.ctypes.data)
    return r.reshape(x.shape)
