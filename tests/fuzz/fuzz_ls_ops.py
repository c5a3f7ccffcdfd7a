#!/usr/bin/env python3
"""tests/fuzz/fuzz_ls_ops.py -- randomized differential test of the other move operators against the C oracle.

    python tests/fuzz/fuzz_ls_ops.py [--seconds 120] [--seed 1] [--max-n 260]

Cases: one operator call (two-opt variant / relocate; all moves or a fixed row per tour; plain or guided
weights with random penalty matrices), or the local_search loop; uniform / integer-grid (many equal deltas) /
clustered points.  Tours and deltas (rounds) must be identical.  One JSON line; exit 1 on any mismatch.
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ml4tspbench_amd as amd
from oracle import ls_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=260)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
t_end = time.time() + a.seconds
cases = fails = 0
stats = {}
last = time.time()
while time.time() < t_end:
    bands = [(3, 12), (12, 90), (90, a.max_n + 1)]
    lo, hi = bands[int(rng.integers(0, len(bands)))]
    hi = min(hi, a.max_n + 1)
    n = int(rng.integers(min(lo, hi - 1), hi))
    inst, tpi = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    layout = rng.choice(["uniform", "grid", "clustered"])
    if layout == "grid":
        pts = rng.integers(0, max(2, int(np.sqrt(n))), size=(inst, n, 2)).astype(np.float64)
    elif layout == "clustered":
        pts = 0.5 + 1e-3 * rng.standard_normal((inst, n, 2))
    else:
        pts = rng.random((inst, n, 2))
    T = inst * tpi
    tours = np.zeros((T, n + 1), dtype=np.int64)
    for k in range(T):
        tours[k, 1:n] = rng.permutation(np.arange(1, n))
    kind = rng.choice(["step", "step_fixed", "step_guided", "step_guided_fixed", "loop"])
    d_p, d_t = torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev)
    if kind == "loop":
        cap = int(rng.choice([1 << 30, int(rng.integers(1, 6))]))
        got_t, got_r = amd.local_search_device(d_p, d_t, tours_per_instance=tpi, max_rounds=cap)
        ref_t, ref_r = ls_oracle.local_search(pts, tours, tours_per_instance=tpi, max_rounds=cap)
        ok = np.array_equal(got_t.cpu().numpy(), ref_t) and np.array_equal(got_r.cpu().numpy(), ref_r)
    else:
        op = int(rng.integers(0, 2))
        fixed = rng.integers(0, n + 1, size=T).astype(np.int32) if "fixed" in kind else None
        gk = pen = None
        if "guided" in kind:
            gk = rng.random(T) * 0.3
            pen = rng.integers(0, 4, size=(T, n, n)).astype(np.float64)
        got_t, got_d = amd.ls_step_device(d_p, d_t, op, tours_per_instance=tpi,
                                          fixed_i=None if fixed is None else torch.from_numpy(fixed).to(dev),
                                          guide_k=None if gk is None else torch.from_numpy(gk).to(dev),
                                          penalty=None if pen is None else torch.from_numpy(pen).to(dev))
        ref_t, ref_d = ls_oracle.step(pts, tours, op, tours_per_instance=tpi, fixed_i=fixed, guide_k=gk, penalty=pen)
        ok = np.array_equal(got_t.cpu().numpy(), ref_t) and np.array_equal(got_d.cpu().numpy(), ref_d)
    key = f"{kind}/{layout}"
    stats[key] = stats.get(key, 0) + 1
    cases += 1
    if not ok:
        fails += 1
        print(json.dumps({"mismatch": {"n": n, "inst": inst, "tpi": tpi, "layout": str(layout), "kind": str(kind),
                                       "case": cases, "seed": a.seed}}), flush=True)
    if time.time() - last > 30:
        print(f"... {cases} cases, {fails} mismatches", file=sys.stderr, flush=True)
        last = time.time()
print(json.dumps({"cases": cases, "mismatches": fails, "seed": a.seed, "by_kind": stats}))
sys.exit(1 if fails else 0)
