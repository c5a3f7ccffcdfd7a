#!/usr/bin/env python3
"""tests/fuzz/fuzz_greedy.py -- randomized differential test of the greedy decoders against the CPU oracle.

    python tests/fuzz/fuzz_greedy.py [--seconds 120] [--seed 1] [--max-n 700]

Every case draws a size, a dtype, a point layout (uniform / grid / clustered / with coincident points) and a
heat-map kind (dense random, distance-proportional with tiny noise = near-ties everywhere, thresholded with
exact zeros, kNN lists with duplicates), decodes it with the dense HIP decoder -- and, for list inputs, with
the sparse one -- and requires the oracle's tours.  Prints one JSON line; exit status 1 on any mismatch.
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ml4tspbench_amd as amd
from oracle import greedy_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=700)
ap.add_argument("--only-case", type=int, default=0, help="replay the random stream and run this case alone, verbosely")
ap.add_argument("--dump", default="", help="with --only-case: save the case's inputs to this .npz")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
t_end = time.time() + a.seconds
cases = fails = 0
kinds = {}
last = time.time()
while time.time() < t_end or (a.only_case and cases < a.only_case):
    bands = [(2, 20), (20, 130), (130, 520), (520, a.max_n + 1)]
    lo, hi = bands[int(rng.integers(0, len(bands)))]
    hi = min(hi, a.max_n + 1)
    n = int(rng.integers(min(lo, hi - 1), hi))
    inst, mpi = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    layout = rng.choice(["uniform", "grid", "clustered", "coincident"])
    if layout == "grid":
        g = int(np.ceil(np.sqrt(n)))
        base = np.stack(np.meshgrid(np.arange(g), np.arange(g)), -1).reshape(-1, 2)[:n] / g
        pts = np.stack([base[rng.permutation(n)] for _ in range(inst)])
    elif layout == "clustered":
        pts = 0.5 + 0.01 * rng.standard_normal((inst, n, 2))
    else:
        pts = rng.random((inst, n, 2))
        if layout == "coincident" and n > 4:
            pts[:, 1] = pts[:, 3]
    d = np.sqrt(((pts[:, :, None] - pts[:, None]) ** 2).sum(-1))
    kind = rng.choice(["dense", "near_ties", "thresholded", "knn"])
    key = f"{kind}/{layout}"
    kinds[key] = kinds.get(key, 0) + 1
    B = inst * mpi
    ok = True
    if kind == "knn" and n >= 4:
        k = int(rng.integers(2, min(n, 24)))
        eis, vals = [], []
        for q in range(inst):
            nn = np.argsort(d[q], axis=1)[:, :k]
            e0, e1 = np.repeat(np.arange(n), k), nn.reshape(-1)
            dup = rng.integers(0, len(e0), size=int(rng.integers(0, 6)))
            pad = rng.integers(0, n, size=(2, 6 - len(dup)))               # equal list lengths per instance
            e0 = np.concatenate([e0, e0[dup], pad[0]]); e1 = np.concatenate([e1, e1[dup], pad[1]])
            eis.append(np.stack([e0, e1]).astype(np.int64))
        ei = np.stack(eis)
        vals = (rng.random((B, ei.shape[2])) + 0.01).astype(dtype)
        if rng.random() < 0.3:
            vals[:, rng.integers(0, vals.shape[1], size=5)] = 0
        # twin points whose pair has no positive sum: the key is -0.0 / 0 = NaN and sorts last (a generator of its
        # own, so that the main stream -- and with it every recorded seed -- stays what it was)
        if layout == "coincident" and np.random.default_rng([a.seed, cases]).random() < 0.5:
            for q in range(inst):
                twin = (d[q][ei[q, 0], ei[q, 1]] == 0) & (ei[q, 0] != ei[q, 1])
                vals[q * mpi:(q + 1) * mpi, twin] = 0
        if a.only_case and cases + 1 != a.only_case:
            cases += 1
            continue
        dense = np.stack([greedy_oracle.densify(vals[m:m + 1], ei[m // mpi], n)[0] for m in range(B)])
        ref = np.concatenate([greedy_oracle.greedy(pts[m // mpi], dense[m:m + 1]) for m in range(B)])
        got_s = amd.greedy_decode_sparse_device(torch.from_numpy(vals).to(dev), torch.from_numpy(ei).to(dev),
                                                torch.from_numpy(pts).to(dev), maps_per_instance=mpi).cpu().numpy()
        ok &= bool(np.array_equal(got_s, ref))
        if a.only_case:
            print("sparse decoder == oracle:", bool(np.array_equal(got_s, ref)), "maps that differ:",
                  [int(m) for m in range(B) if not np.array_equal(got_s[m], ref[m])])
            if a.dump:
                np.savez(a.dump, pts=pts, vals=vals, ei=ei, ref=ref, got_sparse=got_s, mpi=mpi)
        heat = dense
    else:
        if kind == "near_ties":
            eps = 10.0 ** rng.uniform(-15, -4.5)
            heat = (0.4 * np.repeat(d, mpi, axis=0) * (1 + eps * rng.standard_normal((B, n, n)))).astype(dtype)
        else:
            heat = rng.random((B, n, n)).astype(dtype)
            if kind == "thresholded":
                heat[heat < rng.uniform(0.3, 0.95)] = 0
        if a.only_case and cases + 1 != a.only_case:
            cases += 1
            continue
        ref = np.concatenate([greedy_oracle.greedy(pts[m // mpi], heat[m:m + 1]) for m in range(B)])
    got = amd.greedy_decode_device(torch.from_numpy(np.ascontiguousarray(heat)).to(dev), torch.from_numpy(pts).to(dev),
                                   maps_per_instance=mpi).cpu().numpy()
    ok &= bool(np.array_equal(got, ref))
    if a.only_case:
        print("dense decoder == oracle:", bool(np.array_equal(got, ref)), "maps that differ:",
              [int(m) for m in range(B) if not np.array_equal(got[m], ref[m])])
        if a.dump and kind != "knn":
            np.savez(a.dump, pts=pts, heat=heat, ref=ref, got=got, mpi=mpi)
    cases += 1
    if not ok:
        fails += 1
        print(json.dumps({"mismatch": {"n": n, "inst": inst, "mpi": mpi, "dtype": str(np.dtype(dtype)), "layout": str(layout),
                                       "kind": str(kind), "case": cases, "seed": a.seed}}), flush=True)
    if a.only_case and cases >= a.only_case:
        break
    if time.time() - last > 30:
        print(f"... {cases} cases, {fails} mismatches", file=sys.stderr, flush=True)
        last = time.time()
print(json.dumps({"cases": cases, "mismatches": fails, "seed": a.seed, "by_kind": kinds}))
sys.exit(1 if fails else 0)
