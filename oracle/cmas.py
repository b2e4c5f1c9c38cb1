"""ctypes loader of oracle/libmas_oracle.so (C restatement of the alignment search) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmas_oracle.so")
_lib = None


def available():
    return os.path.exists(_SO)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_SO)
        _lib.mas_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.viterbi_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def mas(log_p):
    lp = np.ascontiguousarray(log_p, dtype=np.float32)
    T, N = lp.shape
    path = np.empty(T, dtype=np.int64)
    rc = _load().mas_oracle(lp.ctypes.data, T, N, N, path.ctypes.data)
    assert rc == 0
    return path


def viterbi_decode(log_p, text_lens, feat_lens):
    lp = np.ascontiguousarray(log_p, dtype=np.float32)
    B, Tf, Tx = lp.shape
    tl = np.ascontiguousarray(text_lens, dtype=np.int64)
    fl = np.ascontiguousarray(feat_lens, dtype=np.int64)
    ds = np.empty((B, Tx), dtype=np.float32)
    bl = ctypes.c_double(0)
    rc = _load().viterbi_oracle(lp.ctypes.data, B, Tf, Tx, tl.ctypes.data, fl.ctypes.data, ds.ctypes.data, ctypes.byref(bl))
    assert rc == 0
    return ds, bl.value
