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

# ---- guided local search (search/local_search.py:157-232): whole rounds on the device, all tours side by side
gn, gt_, grounds, gpm = a.n, min(256, a.tours), 5, 20
rng = np.random.default_rng(4321)
gp = rng.random((gt_, gn, 2))
gtours = np.zeros((gt_, gn + 1), dtype=np.int64)
for k in range(gt_):
    gtours[k, 1:gn] = rng.permutation(np.arange(1, gn))
d = np.linalg.norm(gp[:, :, None] - gp[:, None], axis=-1)
gheat = -np.exp(-d / (0.35 / np.sqrt(gn))) * (0.6 + 0.8 * rng.random((gt_, gn, gn)))     # regrets = -heat > 0
g_p, g_t, g_h = torch.from_numpy(gp).to(dev), torch.from_numpy(gtours).to(dev), torch.from_numpy(gheat).to(dev)
amd.guided_ls_device(g_p, g_t, g_h, perturbation_moves=gpm, max_rounds=1); torch.cuda.synchronize()
t0 = time.perf_counter()
gbest = amd.guided_ls_device(g_p, g_t, g_h, perturbation_moves=gpm, max_rounds=grounds)
torch.cuda.synchronize()
g_s = time.perf_counter() - t0
kk = min(8, gt_)
t0 = time.perf_counter()
gref = [ls_oracle.guided_local_search(gp[k], gtours[k], gheat[k], rounds=grounds, perturbation_moves=gpm) for k in range(kk)]
gcpu_s = time.perf_counter() - t0
print(json.dumps({
    "op": "guided local search (tsp_gls), whole rounds on the device", "nodes": gn, "tours": gt_, "rounds": grounds,
    "perturbation_moves": gpm, "gpu_ms": g_s * 1e3, "gpu_rounds_per_s": gt_ * grounds / g_s,
    "cpu_oracle": {"tours": kk, "s": gcpu_s, "rounds_per_s": kk * grounds / gcpu_s, "threads": 1,
                   "note": "oracle/ls_oracle.py: the C operators driven by a Python loop like the reference's"},
    "matches_oracle_on_sample": bool(all(np.array_equal(gbest[k].cpu().numpy(), gref[k]) for k in range(kk))),
}))
