"""oracle/two_opt_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of ``oracle/two_opt_oracle.c``, the scalar CPU restatement of the
reference's best-improvement 2-opt (reference: ``search/two_opt.py:5-51`` and
``ml4tsp-library/ml4tsp/nar/local_search/two_opt.py:18-67``).  Parity status: PINNED
against ``tests/golden/*.npz`` (made by ``oracle/gen_golden.py`` from the reference
itself).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtwo_opt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "two_opt_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtwo_opt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_two_opt.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.oracle_two_opt.restype = ctypes.c_int
        L.oracle_tour_lengths.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.oracle_tour_lengths.restype = ctypes.c_int
        L.oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _dtype_code(points: np.ndarray) -> int:
    if points.dtype == np.float64:
        return 1
    if points.dtype == np.float32:
        return 0
    raise TypeError(f"oracle supports float32/float64 points, got {points.dtype}")


def solve(points: np.ndarray, tours: np.ndarray, *, tours_per_instance: int = 1,
          group_size: int = 1, max_iterations: int = 5000, threads: int = 1):
    """Run the oracle on a batch.

    points (I, N, 2) float32/float64, tours (T, N+1) integer.  Returns
    ``(tours_out int64 (T, N+1), iterations int32 (T,), sweeps int32 (T,))``.
    Arithmetic runs in the dtype of ``points`` exactly like the reference does.
    """
    points = np.ascontiguousarray(points)
    if points.ndim == 2:
        points = points[None]
    out = np.ascontiguousarray(np.array(tours, dtype=np.int64, copy=True))
    if out.ndim == 1:
        out = out[None]
    T, n1 = out.shape
    n = n1 - 1
    assert points.shape[1] == n and points.shape[2] == 2
    assert T <= points.shape[0] * tours_per_instance
    iters = np.zeros(T, dtype=np.int32)
    sweeps = np.zeros(T, dtype=np.int32)
    rc = lib().oracle_two_opt(points.ctypes.data, out.ctypes.data, iters.ctypes.data,
                              sweeps.ctypes.data, T, n, tours_per_instance, group_size,
                              int(max_iterations), _dtype_code(points), int(threads))
    if rc != 0:
        raise ValueError(f"oracle_two_opt failed with code {rc}")
    return out, iters, sweeps


def batched_two_opt(points: np.ndarray, tour: np.ndarray, max_iterations: int = 5000):
    """Same call shape as the reference's ``batched_two_opt_torch`` (search/two_opt.py:5-51):
    one (N,2) coordinate set, a (N+1,) or (B,N+1) batch that stops together.
    Returns ``(tours (B,N+1) int64, iterator)``."""
    tour = np.asarray(tour)
    t2 = tour[None] if tour.ndim == 1 else tour
    out, iters, _ = solve(points[None], t2, tours_per_instance=t2.shape[0],
                          group_size=t2.shape[0], max_iterations=max_iterations)
    return out, int(iters[0]) if len(iters) else 0


def tour_lengths(points: np.ndarray, tours: np.ndarray, tours_per_instance: int = 1) -> np.ndarray:
    points = np.ascontiguousarray(points)
    if points.ndim == 2:
        points = points[None]
    tours = np.ascontiguousarray(np.asarray(tours, dtype=np.int64))
    if tours.ndim == 1:
        tours = tours[None]
    T, n1 = tours.shape
    out = np.zeros(T, dtype=np.float64)
    rc = lib().oracle_tour_lengths(points.ctypes.data, tours.ctypes.data, out.ctypes.data,
                                   T, n1 - 1, tours_per_instance, _dtype_code(points))
    if rc != 0:
        raise ValueError(f"oracle_tour_lengths failed with code {rc}")
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())
