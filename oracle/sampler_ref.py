"""ctypes loader of oracle/sampler_ref.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsage_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "sampler_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.sage_ref_sample_neighbors.restype = ctypes.c_int
    return _lib


def philox(counter, key):
    c = (ctypes.c_uint32 * 4)(*counter)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    _load().sage_ref_philox(c, k, o)
    return list(o)


def sample_neighbors(rowptr, col, nodes, k, seed, tag):
    """-> (nbr int32 [n,k] padded with -1, cnt int32 [n]); include/sage355.h semantics."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    nodes = np.ascontiguousarray(nodes, dtype=np.int32)
    n = nodes.shape[0]
    nbr = np.empty((n, k), dtype=np.int32)
    cnt = np.empty(n, dtype=np.int32)
    vp = ctypes.c_void_p
    rc = _load().sage_ref_sample_neighbors(vp(rowptr.ctypes.data), vp(col.ctypes.data), vp(nodes.ctypes.data),
                                           ctypes.c_int32(n), ctypes.c_int32(k), ctypes.c_uint64(seed),
                                           ctypes.c_uint32(tag), vp(nbr.ctypes.data), vp(cnt.ctypes.data))
    if rc != 0:
        raise ValueError("sage_ref_sample_neighbors: bad k")
    return nbr, cnt
