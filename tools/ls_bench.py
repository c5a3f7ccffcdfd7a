#!/usr/bin/env python3
"""tools/ls_bench.py -- the other move operators (row f4) timed on one MI355X against the CPU oracle.

    python tools/ls_bench.py [--n 100] [--tours 2048]

local_search (two-opt variant + relocate until no improvement, search/local_search.py:125-134) for a
batch of random tours: HIP (tsp2opt_local_search) vs the scalar C oracle with OpenMP, results compared
bit-for-bit on the oracle's sample.  Candidate evaluations are counted as the reference scans them:
per round (N-2)(N-3)/2 two-opt pairs + (N-1)(N-2) - (N-2) relocate pairs."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4tspbench_amd as amd
from oracle import ls_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--tours", type=int, default=2048)
ap.add_argument("--cpu-tours", type=int, default=64)
a = ap.parse_args()
rng = np.random.default_rng(1234)
pts = rng.random((a.tours, a.n, 2))
tours = np.zeros((a.tours, a.n + 1), dtype=np.int64)
for k in range(a.tours):
    tours[k, 1:a.n] = rng.permutation(np.arange(1, a.n))
dev = torch.device("cuda", 0)
d_p, d_t = torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev)
amd.local_search_device(d_p, d_t); torch.cuda.synchronize()
t0 = time.perf_counter()
out, rounds = amd.local_search_device(d_p, d_t)
torch.cuda.synchronize()
gpu_s = time.perf_counter() - t0
per_round = (a.n - 2) * (a.n - 3) // 2 + (a.n - 1) * (a.n - 2) - (a.n - 2)
cands = float(rounds.double().sum().item()) * per_round
k = min(a.cpu_tours, a.tours)
t0 = time.perf_counter()
ref, ref_rounds = ls_oracle.local_search(pts[:k], tours[:k])
cpu_s = time.perf_counter() - t0
print(json.dumps({
    "op": "local_search (two-opt variant + relocate)", "nodes": a.n, "tours": a.tours,
    "gpu_ms": gpu_s * 1e3, "mean_rounds": float(rounds.double().mean().item()),
    "gpu_candidate_moves_per_s": cands / gpu_s, "gpu_tours_per_s": a.tours / gpu_s,
    "cpu_oracle": {"tours": k, "s": cpu_s, "tours_per_s": k / cpu_s, "threads": os.cpu_count()},
    "matches_oracle_on_sample": bool(np.array_equal(out[:k].cpu().numpy(), ref)
                                     and np.array_equal(rounds[:k].cpu().numpy(), ref_rounds)),
}))
