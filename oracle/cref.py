"""ctypes access to the oracle's C restatements (oracle/csrc, built by oracle/Makefile) — test infrastructure."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libpairwise_ref.so")


def available() -> bool:
    return os.path.exists(_PATH)


def _lib():
    lib = ctypes.CDLL(_PATH)
    f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    lib.oracle_pairwise_euclidean_f32.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, f32p]
    lib.oracle_sqdist_f32.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, f32p]
    return lib


def pairwise_euclidean(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((x.shape[0], x.shape[0]), dtype=np.float32)
    _lib().oracle_pairwise_euclidean_f32(x, x.shape[0], x.shape[1], out)
    return out


def sqdist(x: np.ndarray, q_begin: int, q_end: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((q_end - q_begin, x.shape[0]), dtype=np.float32)
    _lib().oracle_sqdist_f32(x, x.shape[0], x.shape[1], q_begin, q_end, out)
    return out
