"""oracle/gen_golden_greedy.py -- TEST INFRASTRUCTURE: golden vectors for the greedy edge-merge
decoder.  Build container only.

Runs the reference's own ``tsp_greedy`` (``search/greedy_search.py:19-74``) with its own Cython
merge (``search/cython_merge/sources/cython_merge.pyx``, compiled by
``oracle/build_ref_cython_merge.sh`` into oracle/_ref/).  ``greedy_search.py`` is loaded by file
path with a stand-in ``search.cython_merge`` package entry that points at that compiled module (the
``search`` package __init__ itself pulls ctypes libs and ml4co_kit and is not imported).

Fixtures (tests/golden/greedy_*.npz): points (N,2) f64, heatmap (P,N,N) f32 [dense] or values
(P,E) + edge_index (2,E) [sparse], tours (P,N+1) int64 -- the reference output.
Scores are drawn from a continuous distribution, so no two different edges tie (numpy's unstable
argsort leaves the order of tied scores undefined: parity for tied scores is unpinned).
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def load_reference_greedy():
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    import cython_merge                                   # compiled from the reference's .pyx
    pkg = types.ModuleType("search")
    sub = types.ModuleType("search.cython_merge")
    sub.merge_cython = cython_merge.merge_cython
    pkg.cython_merge = sub
    sys.modules["search"] = pkg
    sys.modules["search.cython_merge"] = sub
    spec = importlib.util.spec_from_file_location("ref_greedy_search", "/root/reference/search/greedy_search.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_greedy()
    rng = np.random.default_rng(2024)
    os.makedirs(OUT, exist_ok=True)
    for name, n, p, kind in [("n12_p1", 12, 1, "dense"), ("n20_p3", 20, 3, "dense"), ("n50_p4", 50, 4, "dense"),
                             ("n100_p2", 100, 2, "dense"), ("n200_p1", 200, 1, "dense"),
                             ("n500_p1", 500, 1, "dense"), ("n50_peaked_p2", 50, 2, "peaked"),
                             ("n100_sparse_p2", 100, 2, "sparse")]:
        pts = rng.random((n, 2))
        if kind == "sparse":
            k = 10
            d = np.linalg.norm(pts[:, None] - pts[None], axis=-1)
            nbr = np.argsort(d, axis=1)[:, 1:k + 1]
            edge_index = np.stack([np.repeat(np.arange(n), k), nbr.reshape(-1)]).astype(np.int64)
            vals = rng.random((p, edge_index.shape[1])).astype(np.float32) + 1e-3
            tours = ref.tsp_greedy(vals.reshape(-1), pts, edge_index_np=edge_index, sparse_graph=True, parallel_sampling=p)  # 1-D (P*E,), split per sample
            np.savez_compressed(os.path.join(OUT, f"greedy_{name}.npz"), points=pts, values=vals,
                                edge_index=edge_index, tours=np.array(tours, dtype=np.int64))
        else:
            heat = rng.random((p, n, n)).astype(np.float32)
            if kind == "peaked":            # a heatmap that concentrates on a good tour, like a trained model
                d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + np.eye(n)
                heat = (np.exp(-d / 0.05)[None] * (0.5 + rng.random((p, n, n)))).astype(np.float32)
            tours = ref.tsp_greedy(heat, pts, parallel_sampling=p)
            np.savez_compressed(os.path.join(OUT, f"greedy_{name}.npz"), points=pts, heatmap=heat,
                                tours=np.array(tours, dtype=np.int64))
        t = np.array(tours)
        assert t.shape == (p, n + 1) and all(sorted(r[:-1]) == list(range(n)) for r in t)
        print("wrote", name, t.shape, flush=True)


if __name__ == "__main__":
    main()
