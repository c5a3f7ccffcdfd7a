"""oracle/gen_golden_eval.py -- TEST INFRASTRUCTURE: makes tests/golden/eval_*.npz and
tests/golden/txt_format_lines.txt.

Runs ONLY in the build container, where /root/reference exists.

  * Tour costs come from the reference's own ``TSPEvaluator`` (``utils/evaluator.py:5-13``), loaded by
    file path (it imports scipy only): ``dist_mat = scipy.spatial.distance_matrix(points, points)`` and
    a Python loop over the tour's edges.
  * The validity verdicts restate ``ML4TSPNARLocalSearch.is_valid_tour`` + the closing-node check
    (``ml4tsp-library/ml4tsp/nar/local_search/base.py:31-32,67-72``: ``sorted(tour[:-1]) ==
    [0..N-1]`` and ``tour[-1] == 0``); that module needs ``ml4co_kit`` (absent) and cannot be imported.
  * Best-of-P is ``np.argmin`` over an instance's P costs (``models/modelzoo/gnn/tsp_gnn.py:218-221``
    takes ``np.min``; the first minimal tour is the one it belongs to).
  * The text lines follow the dataset writer ``generate.py:147-153`` (``x y x y ... output t1 ... tN t1 ``
    with 1-based nodes and a TRAILING SPACE after the closing node) and ``TSPSolver.to_txt``
    (``solvers/solver.py:129-136``, no closing node, no trailing space).  Neither file can be imported
    (lkh / pyconcorde / tsplib95 at import time), so the two write sequences are restated here on
    seeded data; the stored lines are data, and the parsed arrays are stored next to them.

    python oracle/gen_golden_eval.py
"""
from __future__ import annotations

import importlib.util
import os

import numpy as np

REF = "/root/reference/utils/evaluator.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_evaluator():
    spec = importlib.util.spec_from_file_location("ref_evaluator", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.TSPEvaluator


def random_tour(rng, n):
    t = np.zeros(n + 1, dtype=np.int64)
    t[1:n] = rng.permutation(np.arange(1, n))
    return t


def eval_case(name, Evaluator, rng, inst, tpi, n, dtype=np.float64, scale=1.0, offset=0.0):
    pts = (rng.random((inst, n, 2)) * scale + offset).astype(dtype)
    tours = np.array([random_tour(rng, n) for _ in range(inst * tpi)])
    # invalid tours of every kind the reference's checks catch (indices stay in range: the evaluator
    # itself would raise IndexError otherwise)
    if inst * tpi >= 6 and n >= 8:
        tours[1, 3] = tours[1, 4]                           # duplicate node -> not a permutation
        tours[2, -1] = 3                                    # does not end at node 0
        tours[4] = np.roll(tours[4][:-1], 2).tolist() + [np.roll(tours[4][:-1], 2)[0]]   # closed, starts elsewhere
        tours[5, n // 2] = 0                                # node 0 twice
    costs = np.zeros(len(tours))
    for k, t in enumerate(tours):
        ev = Evaluator(pts[k // tpi])                       # evaluator.py:6-7
        costs[k] = ev.evaluate(t)                           # evaluator.py:9-13
    perm = np.array([sorted(t[:-1]) == [i for i in range(n)] for t in tours])      # base.py:31-32
    ends0 = tours[:, -1] == 0                                                        # base.py:71
    best = np.arange(inst) * tpi + costs.reshape(inst, tpi).argmin(axis=1)           # tsp_gnn.py:220-221
    np.savez_compressed(os.path.join(OUT, name + ".npz"), points=pts, tours=tours, costs=costs,
                        is_permutation=perm, ends_at_zero=ends0, best_index=best,
                        best_cost=costs.reshape(inst, tpi).min(axis=1), tours_per_instance=np.int64(tpi))
    print(f"wrote {name}: T={len(tours)} N={n} invalid={int((~(perm & ends0)).sum())}")


def txt_lines(rng):
    """Three dataset lines exactly as generate.py:147-153 writes them and one as TSPSolver.to_txt does."""
    lines, pts_all, tours_all = [], [], []
    for n in (5, 8, 12):
        batch_nodes_coord = rng.random([1, n, 2])           # generate.py:22-23
        tour = np.append(0, rng.permutation(np.arange(1, n)))   # solver output: N nodes, open
        s = " ".join(str(x) + str(" ") + str(y) for x, y in batch_nodes_coord[0])       # generate.py:148
        s += str(" ") + str("output") + str(" ")                                         # :149
        s += str(" ").join(str(node_idx + 1) for node_idx in tour)                       # :150
        s += str(" ") + str(tour[0] + 1) + str(" ")                                      # :151 (trailing space)
        lines.append(s + "\n")                                                           # :152
        pts_all.append(batch_nodes_coord[0])
        tours_all.append(np.append(tour, tour[0]))
    # TSPSolver.to_txt (solvers/solver.py:129-136): tours as given, no trailing space
    n = 6
    p = rng.random([n, 2])
    tour = np.append(np.append(0, rng.permutation(np.arange(1, n))), 0)
    s = " ".join(str(x) + str(" ") + str(y) for x, y in p)
    s += str(" ") + str("output") + str(" ")
    s += str(" ").join(str(node_idx + 1) for node_idx in tour)
    lines.append(s + "\n")
    with open(os.path.join(OUT, "txt_format_lines.txt"), "w") as fh:
        fh.writelines(lines)
    np.savez_compressed(os.path.join(OUT, "txt_format_parsed.npz"),
                        p0=pts_all[0], p1=pts_all[1], p2=pts_all[2], p3=p,
                        t0=tours_all[0], t1=tours_all[1], t2=tours_all[2], t3=tour)
    print("wrote txt_format_lines.txt (4 lines)")


def main():
    Evaluator = load_evaluator()
    rng = np.random.default_rng(2024)
    eval_case("eval_n20_p4", Evaluator, rng, 6, 4, 20)
    eval_case("eval_n83_p5", Evaluator, rng, 7, 5, 83)
    eval_case("eval_n200_p1", Evaluator, rng, 9, 1, 200)
    eval_case("eval_n64_p8_f32", Evaluator, rng, 3, 8, 64, dtype=np.float32)
    eval_case("eval_n50_p2_offset", Evaluator, rng, 4, 2, 50, scale=1000.0, offset=5.0e5)
    eval_case("eval_n3_p2", Evaluator, rng, 3, 2, 3)
    txt_lines(rng)


if __name__ == "__main__":
    main()
