"""ctypes/numpy front-end to the C length-regulator oracle (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liblr_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.lr_effective_durations.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def effective_durations(ds, ilens, alpha=1.0):
    """ds (B,T) int64, ilens (B,) -> (d_eff (B,T) int64, olens (B,) int64)."""
    ds = np.ascontiguousarray(ds, dtype=np.int64)
    ilens = np.ascontiguousarray(ilens, dtype=np.int32)
    B, T = ds.shape
    d_eff = np.zeros_like(ds)
    olens = np.zeros(B, dtype=np.int64)
    _lib().lr_effective_durations(_p(ds), _p(ilens), B, T, ctypes.c_float(alpha), _p(d_eff), _p(olens))
    return d_eff, olens


def frame_index(d):
    d = np.ascontiguousarray(d, dtype=np.int64)
    idx = np.zeros(int(d.sum()), dtype=np.int64)
    _lib().lr_frame_index(_p(d), d.shape[0], _p(idx))
    return idx


def gather(x, ds, ilens, alpha=1.0):
    """x (B,T,D) f32 -> (out (B,Tmax,D) f32 zero-padded, olens)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T, D = x.shape
    d_eff, olens = effective_durations(ds, ilens, alpha)
    tmax = int(olens.max()) if B else 0
    out = np.empty((B, tmax, D), dtype=np.float32)
    _lib().lr_gather_f32(_p(x), _p(d_eff), _p(olens), B, T, D, ctypes.c_int64(tmax), _p(out))
    return out, olens
