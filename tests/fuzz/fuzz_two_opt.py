#!/usr/bin/env python3
"""tests/fuzz/fuzz_two_opt.py -- randomized differential test of tsp2opt_solve against the CPU oracle.

    python tests/fuzz/fuzz_two_opt.py [--seconds 180] [--seed 1] [--max-n 900]

Every case draws a size, a dtype, a point layout (uniform / integer grid / clustered / duplicates / offset),
start tours (random, or identity = many equal deltas on a grid), a termination mode (independent, or groups
of P tours that stop together), an iteration cap and a kernel choice (automatic, one workgroup per tour,
2-4 workgroups per tour, coupled calls replayed in the lock-step kernel), and requires the oracle's tours,
iteration counts and sweep counts bit for bit.  One JSON line; exit status 1 on any mismatch.
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ml4tspbench_amd as amd
from ml4tspbench_amd import _lib
from oracle import two_opt_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=180)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=900)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
L = _lib.lib()
t_end = time.time() + a.seconds
cases = fails = 0
stats = {}
last = time.time()
try:
    while time.time() < t_end:
        bands = [(2, 12), (12, 130), (130, 520), (520, a.max_n + 1)]
        lo, hi = bands[int(rng.integers(0, len(bands)))]
        hi = min(hi, a.max_n + 1)
        n = int(rng.integers(min(lo, hi - 1), hi))
        dtype = np.float64 if rng.random() < 0.6 else np.float32
        layout = rng.choice(["uniform", "grid", "clustered", "duplicates", "offset"])
        inst = int(rng.integers(1, 4))
        tpi = int(rng.integers(1, 5))
        if n > 1500:                    # oracle time
            inst, tpi = min(inst, 2), min(tpi, 2)
        if layout == "grid":
            pts = rng.integers(0, max(2, int(np.sqrt(n))), size=(inst, n, 2)).astype(np.float64)
        elif layout == "clustered":
            pts = 0.5 + 1e-3 * rng.standard_normal((inst, n, 2))
        elif layout == "offset":
            pts = 1000.0 + rng.random((inst, n, 2))
        else:
            pts = rng.random((inst, n, 2))
            if layout == "duplicates" and n > 6:
                pts[:, 2] = pts[:, 5]
                pts[:, 3] = pts[:, 5]
        pts = pts.astype(dtype)
        T = inst * tpi
        tours = np.zeros((T, n + 1), dtype=np.int64)
        for k in range(T):
            tours[k, 1:n] = np.arange(1, n) if rng.random() < 0.15 else rng.permutation(np.arange(1, n))
        grouped = rng.random() < 0.4
        group = tpi if grouped else 1
        cap = int(rng.choice([5000, 5000, int(rng.integers(1, 40)), int(rng.integers(40, 400))]))
        # the oracle costs iterations x N^2 / 2 per tour: keep a case near a second of CPU
        if n > 400:
            cap = min(cap, 300)
        if n > 1500:
            cap = min(cap, 60)
        variant = rng.choice(["auto", "one_wg", "cluster2", "cluster4", "lockstep"])
        mode = {"auto": -1, "one_wg": 0, "cluster2": 2, "cluster4": 4, "lockstep": -1}[str(variant)]
        L.tsp2opt_debug_cluster_mode(mode)
        L.tsp2opt_debug_tune(6, 1 if (variant == "lockstep" and grouped) else 0)
        L.tsp2opt_debug_tune(9, int(rng.random() < 0.5))       # small batches: wide geometry or that of large batches
        res = amd.two_opt_device(torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev), tours_per_instance=tpi,
                                 group_size=group, max_iterations=cap, want_lengths=False)
        got_t, got_i, got_s = res.tours.cpu().numpy(), res.iterations.cpu().numpy(), res.sweeps.cpu().numpy()
        ref_t, ref_i, ref_s = two_opt_oracle.solve(pts, tours, tours_per_instance=tpi, group_size=group,
                                                   max_iterations=cap, threads=8)
        ok = np.array_equal(got_t, ref_t) and np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s)
        key = f"{variant}/{'grouped' if grouped else 'indep'}"
        stats[key] = stats.get(key, 0) + 1
        cases += 1
        if not ok:
            fails += 1
            print(json.dumps({"mismatch": {"n": n, "inst": inst, "tpi": tpi, "dtype": str(np.dtype(dtype)),
                                           "layout": str(layout), "grouped": bool(grouped), "cap": cap,
                                           "variant": str(variant), "case": cases, "seed": a.seed}}), flush=True)
        if time.time() - last > 30:
            print(f"... {cases} cases, {fails} mismatches", file=sys.stderr, flush=True)
            last = time.time()
finally:
    L.tsp2opt_debug_cluster_mode(-1)
    L.tsp2opt_debug_tune(6, 0)
    L.tsp2opt_debug_tune(9, 1)
print(json.dumps({"cases": cases, "mismatches": fails, "seed": a.seed, "by_variant": stats}))
sys.exit(1 if fails else 0)
