"""oracle/greedy_oracle.py -- TEST INFRASTRUCTURE: ctypes front end of oracle/greedy_oracle.c, the
CPU restatement of the reference's greedy edge-merge decoder (search/greedy_search.py:19-74,
search/cython_merge/sources/cython_merge.pyx:19-120).  Pinned by tests/golden/greedy_*.npz."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgreedy_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "greedy_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgreedy_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def greedy(points: np.ndarray, heat: np.ndarray) -> np.ndarray:
    """points (N,2), heat (P,N,N) float32/float64 dense -> tours (P,N+1) int64."""
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.oracle_greedy.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 3
        _lib.oracle_greedy.restype = ctypes.c_int
    points = np.ascontiguousarray(points, dtype=np.float64)
    heat = np.ascontiguousarray(heat)
    if heat.ndim == 2:
        heat = heat[None]
    p, n, _ = heat.shape
    code = 1 if heat.dtype == np.float64 else 0
    if code == 0:
        heat = heat.astype(np.float32, copy=False)
    tours = np.zeros((p, n + 1), dtype=np.int64)
    rc = _lib.oracle_greedy(points.ctypes.data, heat.ctypes.data, tours.ctypes.data, p, n, code)
    if rc:
        raise ValueError(f"oracle_greedy failed: {rc}")
    return tours


def densify(values: np.ndarray, edge_index: np.ndarray, n: int) -> np.ndarray:
    """(P,E) sparse heat values -> dense (P,N,N) like the coo_matrix(...).toarray() of
    greedy_search.py:41-47 (duplicate entries are summed)."""
    p = values.shape[0]
    out = np.zeros((p, n, n), dtype=values.dtype)
    for s in range(p):
        np.add.at(out[s], (edge_index[0], edge_index[1]), values[s])
    return out
