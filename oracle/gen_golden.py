"""oracle/gen_golden.py -- TEST INFRASTRUCTURE: makes tests/golden/*.npz.

Runs ONLY in the build container, where /root/reference exists.  It loads the reference's
``search/two_opt.py`` by file path (numpy + torch only; the ``search`` package itself is not
imported because its ``__init__`` pulls Cython/ctypes/ml4co_kit, search/__init__.py:5-27),
calls ``batched_two_opt_torch`` / ``tsp_2opt`` on seeded inputs and stores inputs + outputs.
The fixtures are data (inputs and expected outputs); no reference source is copied.

    python oracle/gen_golden.py [--only G1,G3] [--skip-existing]

Two runs per case (see DESIGN.md "What bit-exact means"):
  * "ieee":   the reference code with ``torch.sqrt`` swapped for an IEEE-754 correctly rounded
              square root (numpy's, i.e. the hardware instruction).  This is what the reference
              computes on a CUDA device (``sqrt.rn``) and on CPU builds without MKL VML, and it is
              the arithmetic contract the oracle and the HIP kernels are held to.
  * "native": the reference exactly as it runs in this container: torch 2.10 CPU routes
              ``torch.sqrt`` through MKL VML (``vdSqrt``/``vsSqrt``, VML_HA), which is NOT
              correctly rounded (0.67 % of random inputs are 1 ulp off, e.g. sqrt(0.5)).
              A 1-ulp change only matters where two deltas tie to the last bit (regular grids,
              the fp-noise moves of the grouped mode), so for almost every fixture the two runs
              agree; ``native_equal`` records it per fixture and tests/ asserts which ones.

Fixture layout (every .npz):
    points      (I, N, 2)  float64 or float32 -- exactly what was handed to the reference
    tours_in    (T, N+1)   int64
    tours_out   (T, N+1)   int64               -- reference output, IEEE sqrt
    iterators   (C,)       int64               -- reference `iterator` per CALL (C calls), IEEE sqrt
    tours_out_native / iterators_native        -- the same from the unpatched run
    native_equal ()        bool
    group_size  ()         int  -- tours per reference call (1: one 1-D call per tour)
    tours_per_instance ()  int
    max_iterations ()      int
    entry       ()         str  -- "tsp_2opt" (casts to fp64, two_opt.py:67) or "batched" (dtype kept)
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import time

import numpy as np

REF = "/root/reference/search/two_opt.py"
SKIP_EXISTING = False
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class ieee_sqrt:
    """Swap torch.sqrt for a correctly rounded one while the reference runs."""
    def __enter__(self):
        import torch
        self._torch, self._orig = torch, torch.sqrt
        torch.sqrt = lambda x: torch.from_numpy(np.sqrt(x.numpy()))
        return self

    def __exit__(self, *exc):
        self._torch.sqrt = self._orig
        return False


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_two_opt", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------- input generators
def uniform_points(rng, inst, n, dtype=np.float64):
    # generate.py:22-23 draws np.random.random([B, N, 2])
    return rng.random((inst, n, 2)).astype(dtype)


def random_tour(rng, n):
    # ml4tsp/nar/decoder/random.py:27-29 style: 0, perm(1..N-1), 0
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


def perturbed(rng, tour, n_moves):
    t = tour.copy()
    n = len(t) - 1
    for _ in range(n_moves):
        i, j = sorted(rng.integers(1, n, size=2))
        t[i:j + 1] = t[i:j + 1][::-1]
    return t


# ---------------------------------------------------------------- reference runners
def run_independent(ref, points, tours, tpi, max_it, entry):
    outs, its = [], []
    for k, t in enumerate(tours):
        p = points[k // tpi]
        if entry == "tsp_2opt":
            # tsp_2opt discards the iterator (two_opt.py:66); call the inner function with the
            # same casts so the fixture keeps it, and check tsp_2opt agrees.
            o, it = ref.batched_two_opt_torch(p.astype("float64"), t.astype("int64"),
                                              max_iterations=max_it, device="cpu")
            o2 = ref.tsp_2opt(p, t, None, device="cpu", max_iterations_2opt=max_it)
            assert o2.ndim == 1 and np.array_equal(o2, o[0])
        else:
            o, it = ref.batched_two_opt_torch(p, t.astype("int64"), max_iterations=max_it,
                                              device="cpu")
        outs.append(o[0])
        its.append(it)
    return np.array(outs, dtype=np.int64), np.array(its, dtype=np.int64)


def run_grouped(ref, points, tours, tpi, group, max_it, entry):
    outs, its = [], []
    for g in range(len(tours) // group):
        p = points[(g * group) // tpi]
        tt = tours[g * group:(g + 1) * group]
        if entry == "tsp_2opt":
            o, it = ref.batched_two_opt_torch(p.astype("float64"), tt.astype("int64"),
                                              max_iterations=max_it, device="cpu")
            o2 = ref.tsp_2opt(p, tt, None, device="cpu", max_iterations_2opt=max_it)
            assert np.array_equal(o2, o)
        else:
            o, it = ref.batched_two_opt_torch(p, tt.astype("int64"), max_iterations=max_it,
                                              device="cpu")
        outs.append(o)
        its.append(it)
    return np.concatenate(outs).astype(np.int64), np.array(its, dtype=np.int64)


def save(name, points, tours_in, tours_out, iterators, nat_out, nat_its, group, tpi, max_it, entry):
    path = os.path.join(OUT, name + ".npz")
    same = bool(np.array_equal(tours_out, nat_out) and np.array_equal(iterators, nat_its))
    np.savez_compressed(path, points=points, tours_in=tours_in.astype(np.int64),
                        tours_out=tours_out, iterators=iterators,
                        tours_out_native=nat_out, iterators_native=nat_its,
                        native_equal=np.bool_(same),
                        group_size=np.int64(group), tours_per_instance=np.int64(tpi),
                        max_iterations=np.int64(max_it), entry=np.array(entry))
    print(f"  wrote {path}  T={len(tours_in)} N={tours_in.shape[1]-1} its={iterators.tolist()[:8]}"
          f"{'...' if len(iterators) > 8 else ''} native_equal={same}", flush=True)


def case(name, ref, points, tours, *, tpi=1, group=1, max_it=5000, entry="tsp_2opt"):
    t0 = time.time()
    tours = np.asarray(tours, dtype=np.int64)
    if SKIP_EXISTING and os.path.exists(os.path.join(OUT, name + ".npz")):
        print(f"  {name}: exists, skipped", flush=True)
        return
    def run():
        if group == 1:
            return run_independent(ref, points, tours, tpi, max_it, entry)
        return run_grouped(ref, points, tours, tpi, group, max_it, entry)
    with ieee_sqrt():
        out, its = run()
    nat_out, nat_its = run()
    save(name, points, tours, out, its, nat_out, nat_its, group, tpi, max_it, entry)
    print(f"  {name}: {time.time()-t0:.1f}s", flush=True)


# ---------------------------------------------------------------- fixture list
def g1(ref):   # N=50, 64 tours, independent, fp64, NN-style and random starts
    rng = np.random.default_rng(101)
    pts = uniform_points(rng, 64, 50)
    tours = [nn_tour(pts[k]) if k % 2 == 0 else random_tour(rng, 50) for k in range(64)]
    case("g1_n50_indep_f64", ref, pts, tours)


def g2(ref):   # N=100, 32 tours
    rng = np.random.default_rng(102)
    pts = uniform_points(rng, 32, 100)
    tours = [perturbed(rng, nn_tour(pts[k]), 6) if k % 2 == 0 else random_tour(rng, 100)
             for k in range(32)]
    case("g2_n100_indep_f64", ref, pts, tours)


def g3(ref):   # grouped (P=16) and the same tours independently; N=50 and N=500
    rng = np.random.default_rng(103)
    pts = uniform_points(rng, 20, 50)
    tours = []
    for k in range(20):
        base = nn_tour(pts[k])
        for s in range(16):
            tours.append(perturbed(rng, base, 1 + s) if k % 2 == 0 else random_tour(rng, 50))
    case("g3_n50_group16_f64", ref, pts, tours, tpi=16, group=16)
    case("g3_n50_group16_as_indep_f64", ref, pts, tours, tpi=16, group=1)
    # a group that is smaller than the instance's tour count (two calls per instance)
    case("g3_n50_group8_tpi16_f64", ref, pts[:6], tours[:96], tpi=16, group=8)

    rng = np.random.default_rng(1031)
    pts = uniform_points(rng, 3, 500)
    tours = []
    for k in range(3):
        base = nn_tour(pts[k])
        for s in range(16):
            tours.append(perturbed(rng, base, 2 + 2 * s) if k < 2 else random_tour(rng, 500))
    case("g3_n500_group16_f64", ref, pts, tours, tpi=16, group=16)
    case("g3_n500_group16_as_indep_f64", ref, pts, tours, tpi=16, group=1)


def g4(ref):   # N=1000, 2 tours
    rng = np.random.default_rng(104)
    pts = uniform_points(rng, 2, 1000)
    tours = [perturbed(rng, nn_tour(pts[0]), 10), perturbed(rng, nn_tour(pts[1]), 40)]
    case("g4_n1000_indep_f64", ref, pts, tours)


def g5(ref):   # N=10000, 1 tour, caps 1 / 3 / 32 (about 4 s and 7 GB per iteration here)
    rng = np.random.default_rng(105)
    pts = uniform_points(rng, 1, 10000)
    t = perturbed(rng, nn_tour(pts[0]), 30)
    for cap in (1, 3, 32):
        case(f"g5_n10000_cap{cap}_f64", ref, pts, [t], max_it=cap)


def g6(ref):   # iteration caps at N=50, independent and grouped
    rng = np.random.default_rng(106)
    pts = uniform_points(rng, 8, 50)
    tours = [random_tour(rng, 50) for _ in range(8)]
    for cap in (1, 7):
        case(f"g6_n50_cap{cap}_f64", ref, pts, tours, max_it=cap)
    pts = uniform_points(rng, 2, 50)
    tours = [random_tour(rng, 50) for _ in range(16)]
    for cap in (1, 7, 20):
        case(f"g6_n50_group8_cap{cap}_f64", ref, pts, tours, tpi=8, group=8, max_it=cap)


def g7(ref):   # fp32 arithmetic: fp32 points handed straight to batched_two_opt_torch
    rng = np.random.default_rng(107)
    pts = uniform_points(rng, 32, 50, np.float32)
    tours = [random_tour(rng, 50) for _ in range(32)]
    case("g7_n50_indep_f32", ref, pts, tours, entry="batched")
    pts = uniform_points(rng, 8, 200, np.float32)
    tours = [perturbed(rng, nn_tour(pts[k]), 8) if k % 2 else random_tour(rng, 200)
             for k in range(8)]
    case("g7_n200_indep_f32", ref, pts, tours, entry="batched")
    pts = uniform_points(rng, 2, 100, np.float32)
    tours = [random_tour(rng, 100) for _ in range(16)]
    case("g7_n100_group8_f32", ref, pts, tours, tpi=8, group=8, entry="batched")
    pts = uniform_points(rng, 2, 500, np.float32)
    tours = [perturbed(rng, nn_tour(pts[k]), 12) for k in range(2)]
    case("g7_n500_indep_f32", ref, pts, tours, entry="batched")


def g8(ref):   # fp32-valued coordinates up-cast to fp64 (solvers/solver.py:77 -> two_opt.py:67)
    rng = np.random.default_rng(108)
    pts = uniform_points(rng, 16, 100, np.float32)
    tours = [random_tour(rng, 100) for _ in range(16)]
    case("g8_n100_f32values_f64", ref, pts, tours, entry="tsp_2opt")


def g9(ref):   # edge cases
    rng = np.random.default_rng(109)
    for n in (2, 3, 4, 5, 6, 7, 9, 10, 11, 13):
        pts = uniform_points(rng, 12, n)
        tours = [random_tour(rng, n) for _ in range(12)]
        case(f"g9_n{n}_indep_f64", ref, pts, tours)
        case(f"g9_n{n}_group4_f64", ref, pts[:3], tours, tpi=4, group=4)
    # duplicate points
    pts = uniform_points(rng, 4, 40)
    pts[:, 10:20] = pts[:, 0:10]
    pts[:, 30:35] = pts[:, 5:6]
    tours = [random_tour(rng, 40) for _ in range(4)]
    case("g9_n40_duplicates_f64", ref, pts, tours)
    # regular grids: many exactly tied deltas
    for side in (5, 8, 12):
        gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64))
        grid = np.stack([gx.ravel(), gy.ravel()], axis=1) / side
        n = side * side
        pts = np.repeat(grid[None], 6, axis=0)
        tours = [random_tour(rng, n) for _ in range(5)] + [np.append(np.arange(n), 0)]
        case(f"g9_grid{side}_indep_f64", ref, pts, tours)
        case(f"g9_grid{side}_group6_f64", ref, pts[:1], tours, tpi=6, group=6)
    gx, gy = np.meshgrid(np.arange(8, dtype=np.float32), np.arange(8, dtype=np.float32))
    grid = (np.stack([gx.ravel(), gy.ravel()], axis=1) / np.float32(8)).astype(np.float32)
    pts = np.repeat(grid[None], 4, axis=0)
    tours = [random_tour(rng, 64) for _ in range(4)]
    case("g9_grid8_indep_f32", ref, pts, tours, entry="batched")
    # collinear points
    pts = np.zeros((4, 30, 2))
    pts[:, :, 0] = rng.random((4, 30))
    tours = [random_tour(rng, 30) for _ in range(4)]
    case("g9_n30_collinear_f64", ref, pts, tours)
    # already 2-optimal input (feed a reference output back in): 0 iterations
    pts = uniform_points(rng, 4, 60)
    tours = [random_tour(rng, 60) for _ in range(4)]
    with ieee_sqrt():
        opt, _ = run_independent(ref, pts, np.array(tours), 1, 5000, "tsp_2opt")
    case("g9_n60_already_optimal_f64", ref, pts, opt)
    # coordinates far from the origin and with a large scale
    pts = uniform_points(rng, 4, 80) * 1000.0 + 5.0e5
    tours = [random_tour(rng, 80) for _ in range(4)]
    case("g9_n80_offset_scaled_f64", ref, pts, tours)
    pts = uniform_points(rng, 4, 80) * 1.0e-3
    case("g9_n80_tiny_scale_f64", ref, pts, tours)
    # clustered instance (generate.py cluster mode is out of scope; this is just a non-uniform set)
    centers = rng.random((5, 2))
    pts = (centers[rng.integers(0, 5, size=(4, 120))] + 0.02 * rng.standard_normal((4, 120, 2)))
    tours = [random_tour(rng, 120) for _ in range(4)]
    case("g9_n120_clustered_f64", ref, pts, tours)
    # int32 tours are accepted by the reference; stored as int64 here
    pts = uniform_points(rng, 2, 33)
    tours = [random_tour(rng, 33) for _ in range(2)]
    case("g9_n33_indep_f64", ref, pts, tours)


def g10(ref):  # coupled groups whose converged tours RE-ACTIVATE: with coordinates of scale 1e-4 many real
    # moves gain less than the 1e-6 threshold, so a tour that has stopped on its own keeps applying them
    # while its siblings run (two_opt.py:40-42) and can be active again when the slowest sibling
    # converges -- the group then runs past the maximum of its tours' own iteration counts
    rng = np.random.default_rng(110)
    for n, inst, scale, dt in ((12, 6, 1.0e-5, np.float64), (50, 8, 1.0e-4, np.float64),
                               (100, 4, 3.0e-5, np.float64), (50, 6, 1.0e-4, np.float32)):
        pts = (uniform_points(rng, inst, n) * scale).astype(dt)
        tours = [random_tour(rng, n) for _ in range(inst * 8)]
        tag = "f64" if dt == np.float64 else "f32"
        entry = "tsp_2opt" if dt == np.float64 else "batched"
        case(f"g10_n{n}_reactivating_group8_{tag}", ref, pts, tours, tpi=8, group=8, entry=entry)
        case(f"g10_n{n}_reactivating_as_indep_{tag}", ref, pts, tours, tpi=8, group=1, entry=entry)
    pts = uniform_points(rng, 4, 50) * 1.0e-4
    tours = [random_tour(rng, 50) for _ in range(32)]
    case("g10_n50_reactivating_group8_cap40_f64", ref, pts, tours, tpi=8, group=8, max_it=40)


ALL = {"G10": g10, "G1": g1, "G2": g2, "G3": g3, "G4": g4, "G5": g5, "G6": g6, "G7": g7, "G8": g8, "G9": g9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-existing", action="store_true")
    args = ap.parse_args()
    global SKIP_EXISTING
    SKIP_EXISTING = args.skip_existing
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    names = [s for s in args.only.split(",") if s] or list(ALL)
    for nm in names:
        print(f"[{nm}]", flush=True)
        ALL[nm](ref)


if __name__ == "__main__":
    sys.exit(main())
