"""oracle/ls_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of ``oracle/ls_oracle.c`` (the reference's two-opt / relocate operators and their
``local_search`` loop, search/local_search.py:9-134, search/relocate.py:5-38) plus a Python restatement
of the guided-local-search driver around them (search/local_search.py:157-232 ``tsp_gls``; the same
loop is ml4tsp/nar/local_search/guided_ls.py:23-88), with a round limit in place of the wall clock.
Parity status: PINNED against tests/golden/ls_*.npz (oracle/gen_golden_ls.py runs the reference).
Only ``tests/`` may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libls_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ls_oracle.c")
    stale = not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libls_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        L.oracle_ls_step.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp]
        L.oracle_ls_step.restype = ctypes.c_int
        L.oracle_local_search.argtypes = [vp, vp, vp, i64, i32, i32, i32]
        L.oracle_local_search.restype = ctypes.c_int
        _lib = L
    return _lib


def _prep(points, tours):
    points = np.ascontiguousarray(np.asarray(points, dtype=np.float64))
    if points.ndim == 2:
        points = points[None]
    out = np.ascontiguousarray(np.array(tours, dtype=np.int64, copy=True))
    if out.ndim == 1:
        out = out[None]
    assert points.shape[1] + 1 == out.shape[1]
    return points, out


def step(points, tours, op: int, *, tours_per_instance: int = 1, fixed_i=None, guide_k=None, penalty=None):
    """One get_two_opt_tour (op 0) / get_relocate_tour (op 1) call per tour.
    Returns ``(tours_out (T, N+1) int64, delta (T,) float64)``."""
    points, out = _prep(points, tours)
    T, n = out.shape[0], points.shape[1]
    delta = np.zeros(T)
    fi = None if fixed_i is None else np.ascontiguousarray(np.asarray(fixed_i, dtype=np.int32).reshape(T))
    gk = None if guide_k is None else np.ascontiguousarray(np.asarray(guide_k, dtype=np.float64).reshape(T))
    pen = None if penalty is None else np.ascontiguousarray(np.asarray(penalty, dtype=np.float64).reshape(T, n, n))
    rc = lib().oracle_ls_step(points.ctypes.data, out.ctypes.data, delta.ctypes.data, T, n, tours_per_instance, op,
                              None if fi is None else fi.ctypes.data, None if gk is None else gk.ctypes.data,
                              None if pen is None else pen.ctypes.data)
    assert rc == 0
    return out, delta


def local_search(points, tours, *, tours_per_instance: int = 1, max_rounds: int = 1 << 30):
    """local_search() per tour.  Returns ``(tours_out, rounds int32)``."""
    points, out = _prep(points, tours)
    T, n = out.shape[0], points.shape[1]
    rounds = np.zeros(T, dtype=np.int32)
    rc = lib().oracle_local_search(points.ctypes.data, out.ctypes.data, rounds.ctypes.data, T, n,
                                   tours_per_instance, int(max_rounds))
    assert rc == 0
    return out, rounds


def tour_cost(points: np.ndarray, tour) -> float:
    """TSPEvaluator.evaluate (utils/evaluator.py:9-13): sequential double sum over the tour's edges."""
    t = np.asarray(tour)
    d = np.sqrt(np.sum(np.abs(points[t[1:]] - points[t[:-1]]) ** 2, axis=-1))
    total = 0
    for x in d:
        total += x
    return total


def guided_local_search(points: np.ndarray, tour, heat: np.ndarray, *, rounds: int, perturbation_moves: int):
    """tsp_gls for one tour with the wall-clock test replaced by a count of outer rounds."""
    points = np.asarray(points, dtype=np.float64)
    n = points.shape[0]
    adj = -1 * np.asarray(heat)
    adj = np.where(adj > 0, adj, 0)
    tour = np.asarray(tour, dtype=np.int64)
    init_cost = tour_cost(points, tour)
    k = 0.1 * init_cost / n
    cur, _ = local_search(points, tour)
    cur = cur[0]
    cur_cost = tour_cost(points, cur)
    best, best_cost = cur.copy(), cur_cost
    pen = np.zeros((n, n))
    for _ in range(rounds):
        moves = 0
        while moves < perturbation_moves:
            max_util, max_e = 0, None
            for src, dst in zip(cur[:-1], cur[1:]):
                if src > dst:
                    src, dst = dst, src
                util = adj[src, dst] / (1 + pen[src, dst])
                if util > max_util or max_e is None:
                    max_util, max_e = util, [int(src), int(dst)]
            src, dst = max_e
            pen[src, dst] += 1.
            pen[dst, src] += 1.
            for node in max_e:
                if node != 0:
                    i = int(np.nonzero(cur == node)[0][0])
                    for op in (0, 1):
                        new, delta = step(points, cur, op, fixed_i=[i], guide_k=[k], penalty=pen[None])
                        if delta[0] < 0:
                            cur = new[0]
                            moves += 1
        cur, _ = local_search(points, cur)
        cur = cur[0]
        cur_cost = tour_cost(points, cur)
        if cur_cost < best_cost:
            best, best_cost = cur.copy(), cur_cost
    return best
