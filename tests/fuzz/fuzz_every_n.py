"""Diagnostic fuzz: every N in a range through the GPU path against the CPU oracle (bit-exact).
    python tests/fuzz/fuzz_every_n.py [lo] [hi] [step] [cluster workgroups, 0 = automatic kernel choice] [variant]
Prints the N that differ (nothing = all equal).  The same loop is in tests/test_gpu_parity.py at a coarser grid.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ml4tspbench_amd as amd
from ml4tspbench_amd import _lib
from oracle import two_opt_oracle as oracle

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 340
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
force = int(sys.argv[4]) if len(sys.argv) > 4 else 0
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
L = _lib.lib()
L.tsp2opt_debug_cluster_mode(force if force else -1)
L.tsp2opt_debug_tune(3, variant)
dev = torch.device("cuda", 0)
bad = []
for n in range(lo, hi + 1, step):
    for dtype in (np.float64, np.float32):
        rng = np.random.default_rng(n * 3 + (dtype == np.float32))
        tours_n = 3
        pts = rng.random((tours_n, n, 2)).astype(dtype)
        tours = np.zeros((tours_n, n + 1), dtype=np.int64)
        for k in range(tours_n):
            tours[k, 1:n] = rng.permutation(np.arange(1, n))
        cap = 5000 if n <= 1200 else 100
        ref, ref_it, ref_sw = oracle.solve(pts, tours, max_iterations=cap, threads=8)
        res = amd.two_opt_device(torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev), max_iterations=cap)
        torch.cuda.synchronize()
        ok = (np.array_equal(res.tours.cpu().numpy(), ref) and np.array_equal(res.iterations.cpu().numpy(), ref_it)
              and int(res.status.cpu().numpy().max()) & (15 | 64) == 0)
        if not ok:
            bad.append((n, dtype.__name__))
            print("MISMATCH", n, dtype.__name__, res.iterations.cpu().numpy(), ref_it, res.status.cpu().numpy(), flush=True)
    if n % 50 == 0:
        print("..", n, flush=True)
print("done", lo, hi, step, force, variant, "bad:", bad)
