"""Diagnostic fuzz of the greedy decoder: a grid of N, dense / sparse (kNN-masked) / tie-heavy heat maps,
against the CPU oracle (bit-exact tours).   python tools/fuzz_greedy.py [lo] [hi] [step]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ml4tspbench_amd import greedy_decode_device
from oracle import greedy_oracle

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 300
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
bad = []
for n in range(lo, hi + 1, step):
    rng = np.random.default_rng(n)
    for kind in ("dense", "sparse", "ties", "grid"):
        for dtype in (np.float32, np.float64):
            inst, mpi = 2, 2
            pts = rng.random((inst, n, 2))
            heat = rng.random((inst * mpi, n, n))
            if kind == "sparse":                       # only a few nonzero entries per row, rest exactly 0
                k = min(n - 1, 5)
                thr = np.sort(heat, axis=2)[:, :, n - k][:, :, None]
                heat = np.where(heat >= thr, heat, 0.0)
            elif kind == "ties":                       # quantised values: many equal keys at equal distances do not
                heat = np.round(heat * 4) / 4          # exist here, but equal heat values stress the list refills
            elif kind == "grid":                       # lattice points + quantised heat: exact key ties
                side = int(np.ceil(np.sqrt(n)))
                g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:n] / side
                pts = np.stack([g, g[::-1]]).astype(np.float64)
                heat = np.round(heat * 2) / 2
            heat = heat.astype(dtype)
            got = greedy_decode_device(torch.from_numpy(heat).to(dev), torch.from_numpy(pts).to(dev),
                                       maps_per_instance=mpi)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            for m in range(inst * mpi):
                ref = greedy_oracle.greedy(pts[m // mpi], heat[m])
                if not np.array_equal(got[m], ref[0]):
                    bad.append((n, kind, dtype.__name__, m))
                    print("MISMATCH", n, kind, dtype.__name__, m, flush=True)
                    break
    if n % 25 == 0:
        print("..", n, flush=True)
print("done", lo, hi, step, "bad:", bad[:20], len(bad))
