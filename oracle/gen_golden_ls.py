"""oracle/gen_golden_ls.py -- TEST INFRASTRUCTURE: makes tests/golden/ls_*.npz (SURVEY.md 8f row f4,
"the other move operators").

Runs ONLY in the build container, where /root/reference exists.  Loads the reference's
``search/relocate.py`` and ``search/local_search.py`` by file path (numpy / scipy / itertools only; the
latter imports ``utils.evaluator``, which resolves to the reference's own file with /root/reference on
sys.path; the ``search`` package itself is not imported, its __init__ pulls Cython / ctypes / ml4co_kit).
The fixtures are data: inputs and the reference's outputs.

  ls_relocate_*   tsp_relocate (one best relocate move per tour, search/relocate.py:40-61)
  ls_step_*       get_two_opt_tour / get_relocate_tour, free and with fixed_i, on plain and on guided
                  weights (dist + k * penalty), with the returned delta
  ls_loop_*       local_search (two-opt + relocate until no improvement) and tsp_ls
  ls_gls_*        tsp_gls with ``time.time`` of the reference module replaced by a counter, so that the
                  wall-clock loop runs a fixed number of rounds (the code is otherwise untouched)

    python oracle/gen_golden_ls.py
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import scipy.spatial

REFROOT = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load(name, rel):
    if REFROOT not in sys.path:
        sys.path.insert(0, REFROOT)
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def random_tour(rng, n):
    t = np.zeros(n + 1, dtype=np.int64)
    t[1:n] = rng.permutation(np.arange(1, n))
    return t


def nn_tour(points):
    n = len(points)
    left = np.ones(n, dtype=bool)
    t = np.zeros(n + 1, dtype=np.int64)
    left[0] = False
    cur = 0
    for k in range(1, n):
        d = np.sum((points - points[cur]) ** 2, axis=1)
        d[~left] = np.inf
        cur = int(np.argmin(d))
        t[k] = cur
        left[cur] = False
    return t


class FakeClock:
    """time.time() stand-in: 0, 1, 2, ... per call (tsp_gls reads it once for the limit and once per
    round), so `time_limit = R` gives exactly R - 1 rounds."""
    def __init__(self):
        self.t = -1

    def time(self):
        self.t += 1
        return float(self.t)


def main():
    rel = load("ref_relocate", "search/relocate.py")
    ls = load("ref_local_search", "search/local_search.py")
    rng = np.random.default_rng(404)

    # ---- tsp_relocate
    for n, inst in ((8, 6), (30, 6), (100, 4)):
        pts = rng.random((inst, n, 2))
        tours = np.array([random_tour(rng, n) if k % 2 else nn_tour(pts[k]) for k in range(inst)])
        out = rel.tsp_relocate(pts, tours, adj_mat=np.zeros((inst, n, n)))
        np.savez_compressed(os.path.join(OUT, f"ls_relocate_n{n}.npz"), points=pts, tours_in=tours,
                            tours_out=np.array(out, dtype=np.int64))
        print(f"ls_relocate_n{n}: changed {int((np.array(out) != tours).any(axis=1).sum())} of {inst}")

    # ---- single operator calls, free / fixed_i, plain / guided
    for n, inst in ((12, 8), (40, 8), (90, 4)):
        pts = rng.random((inst, n, 2))
        tours = np.array([random_tour(rng, n) for _ in range(inst)])
        pen = np.zeros((inst, n, n))
        for k in range(inst):               # a few penalised tour edges, like the GLS perturbation
            for _ in range(5):
                p = int(rng.integers(0, n))
                a, b = int(tours[k, p]), int(tours[k, p + 1])
                pen[k, a, b] += 1.0
                pen[k, b, a] += 1.0
        gk = 0.1 * rng.random(inst) + 0.02
        fixed = rng.integers(1, n, size=inst).astype(np.int32)
        rec = {}
        for op, fn in ((0, ls.get_two_opt_tour), (1, ls.get_relocate_tour)):
            for guided in (0, 1):
                for fx in (0, 1):
                    outs, deltas = [], []
                    for k in range(inst):
                        dm = scipy.spatial.distance_matrix(pts[k], pts[k])
                        w = dm + gk[k] * pen[k] if guided else dm
                        d, t = fn(tours[k].tolist(), w, int(fixed[k]) if fx else None)
                        outs.append(t)
                        deltas.append(d)
                    rec[f"out_op{op}_g{guided}_f{fx}"] = np.array(outs, dtype=np.int64)
                    rec[f"delta_op{op}_g{guided}_f{fx}"] = np.array(deltas, dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, f"ls_step_n{n}.npz"), points=pts, tours_in=tours, penalty=pen,
                            guide_k=gk, fixed_i=fixed, **rec)
        print(f"ls_step_n{n}: moved", {k: int((v < 0).sum()) for k, v in rec.items() if k.startswith("delta")})

    # ---- local_search / tsp_ls
    for n, inst in ((10, 8), (40, 6), (80, 3)):
        pts = rng.random((inst, n, 2))
        tours = np.array([random_tour(rng, n) if k % 2 else nn_tour(pts[k]) for k in range(inst)])
        outs = []
        for k in range(inst):
            dm = scipy.spatial.distance_matrix(pts[k], pts[k])
            o = ls.local_search(tours[k].tolist(), dm)
            # (tsp_ls needs a Python list: with an ndarray tour its relocate() calls ndarray.pop and raises)
            o2 = ls.tsp_ls(pts[k], tours[k].tolist(), np.zeros((n, n)))
            assert list(o2[0]) == list(o)
            outs.append(o)
        np.savez_compressed(os.path.join(OUT, f"ls_loop_n{n}.npz"), points=pts, tours_in=tours,
                            tours_out=np.array(outs, dtype=np.int64))
        print(f"ls_loop_n{n}: done")

    # ---- tsp_gls with a counting clock
    for n, inst, rounds, pm in ((12, 4, 3, 5), (30, 3, 3, 10), (50, 2, 2, 20)):
        pts = rng.random((inst, n, 2))
        tours = np.array([random_tour(rng, n) for _ in range(inst)])
        heat = -rng.random((inst, n, n))            # tsp_gls negates its "adj_mat" and keeps what is positive
        heat = (heat + heat.transpose(0, 2, 1)) / 2
        outs = []
        for k in range(inst):
            ls.time = FakeClock()
            o = ls.tsp_gls(pts[k], tours[k], heat[k], time_limit=rounds + 1, perturbation_moves=pm)
            outs.append(o[0])
        np.savez_compressed(os.path.join(OUT, f"ls_gls_n{n}.npz"), points=pts, tours_in=tours, heat=heat,
                            tours_out=np.array(outs, dtype=np.int64), rounds=np.int64(rounds),
                            perturbation_moves=np.int64(pm))
        print(f"ls_gls_n{n}: done")


if __name__ == "__main__":
    main()
