"""oracle/gen_golden_pipeline.py -- TEST INFRASTRUCTURE: golden vectors for the CHAINED pipeline
heat map -> greedy edge-merge decode -> 2-opt, made by ONE run of the reference.  Build container only.

The reference chains the two steps in its test loop (``models/modelzoo/gnn/tsp_gnn.py:169-210``):
``tours = tsp_greedy(heatmap, np_points, edge_index_np, sparse_graph, parallel_sampling)``
(``search/greedy_search.py:19-74``) and then ``tsp_2opt(np_points, tours, adj_mat, device=...)``
(``search/two_opt.py:54-73``) on the list it got back -- a (P, N+1) batch, i.e. the GROUP-COUPLED
call when parallel_sampling > 1.  This script does exactly that with the reference's own functions
(``tsp_greedy`` with its Cython merge compiled into oracle/_ref/, loaded as in gen_golden_greedy.py;
``two_opt.py`` loaded by file path as in gen_golden.py, ``torch.sqrt`` swapped for the IEEE square root
-- DESIGN.md "What bit-exact means") and stores heat map -> decoded tours -> final tours.

Fixtures (tests/golden/pipe_*.npz): points (N,2) f64, heatmap (P,N,N) f32 [dense] or values (P,E) +
edge_index (2,E) [sparse], decoded (P,N+1) int64, tours_out (P,N+1) int64, iterator () int64 (ieee run),
tours_out_native / iterator_native (the reference as it runs here), native_equal.
"""
import os

import numpy as np

from gen_golden import ieee_sqrt, load_reference
from gen_golden_greedy import load_reference_greedy

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def heat_for(rng, pts, p, noise):
    """A "nearest edges are hot" map times per-sample noise (continuous values: no two edges tie)."""
    n = len(pts)
    d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + np.eye(n)
    tau = 0.35 / np.sqrt(n)
    return (np.exp(-d / tau)[None] * (noise + rng.random((p, n, n)))).astype(np.float32)


def main():
    greedy = load_reference_greedy()
    two_opt = load_reference()
    rng = np.random.default_rng(4242)
    os.makedirs(OUT, exist_ok=True)
    for name, n, p, kind in [("n50_p4", 50, 4, "dense"), ("n100_p2", 100, 2, "dense"),
                             ("n500_p1", 500, 1, "dense"), ("n100_sparse_p2", 100, 2, "sparse")]:
        pts = rng.random((n, 2))
        store = {"points": pts}
        if kind == "sparse":
            # k nearest neighbours, every edge listed with a positive value.  k is large enough for the listed
            # edges alone to close the tour, so no zero-score edge (whose order among ties is undefined in the
            # reference, numpy's unstable argsort) is ever taken: the test checks that.
            k = 30
            d = np.linalg.norm(pts[:, None] - pts[None], axis=-1)
            nbr = np.argsort(d, axis=1)[:, 1:k + 1]
            edge_index = np.stack([np.repeat(np.arange(n), k), nbr.reshape(-1)]).astype(np.int64)
            base = np.exp(-d[edge_index[0], edge_index[1]] / (0.35 / np.sqrt(n)))
            vals = (base[None] * (0.6 + rng.random((p, edge_index.shape[1])))).astype(np.float32)
            decoded = greedy.tsp_greedy(vals.reshape(-1), pts, edge_index_np=edge_index, sparse_graph=True,
                                        parallel_sampling=p)
            store.update(values=vals, edge_index=edge_index)
        else:
            heat = heat_for(rng, pts, p, 0.6)
            decoded = greedy.tsp_greedy(heat, pts, parallel_sampling=p)
            store.update(heatmap=heat)
        dec = np.array(decoded, dtype=np.int64)
        assert dec.shape == (p, n + 1)
        # tsp_gnn.py:204-210: the list of P tours goes to tsp_2opt as one batch; max_iterations_2opt default
        with ieee_sqrt():
            out = two_opt.tsp_2opt(pts, decoded, None, device="cpu")
            _, it = two_opt.batched_two_opt_torch(pts.astype("float64"), dec.astype("int64"), 5000, "cpu")
        nat = two_opt.tsp_2opt(pts, decoded, None, device="cpu")
        _, it_nat = two_opt.batched_two_opt_torch(pts.astype("float64"), dec.astype("int64"), 5000, "cpu")
        out, nat = np.atleast_2d(np.asarray(out)), np.atleast_2d(np.asarray(nat))
        same = bool(np.array_equal(out, nat) and it == it_nat)
        np.savez_compressed(os.path.join(OUT, f"pipe_{name}.npz"), decoded=dec, tours_out=out.astype(np.int64),
                            iterator=np.int64(it), tours_out_native=nat.astype(np.int64),
                            iterator_native=np.int64(it_nat), native_equal=np.bool_(same), **store)
        print(f"wrote pipe_{name}: decoded {dec.shape}, 2-opt iterator {it} (native {it_nat}, equal {same})", flush=True)


if __name__ == "__main__":
    main()
