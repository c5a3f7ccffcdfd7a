#!/usr/bin/env python3
"""tools/nt_probe.py -- threads per workgroup against batch size: random-start tours of N cities, T tours, timed with
tsp2opt_debug_tune(0, nt) for several nt (0 = the planner's choice).  usage: nt_probe.py N T nt [nt ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ml4tspbench_amd as amd
from ml4tspbench_amd import _lib
n, T = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)
pts = torch.from_numpy(rng.random((T, n, 2))).to(dev)
perm = np.stack([np.concatenate([[0], 1 + rng.permutation(n - 1), [0]]) for _ in range(T)])
tours = torch.from_numpy(perm).to(dev)
L = _lib.lib()
for nt in [int(a) for a in sys.argv[3:]]:
    L.tsp2opt_debug_tune(0, nt)
    for _ in range(2):
        res = amd.two_opt_device(pts, tours, tours_per_instance=1, want_lengths=False)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        res = amd.two_opt_device(pts, tours, tours_per_instance=1, want_lengths=False)
    ev[1].record(); torch.cuda.synchronize()
    print(f"N={n} T={T} nt={nt}: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms  iterations mean {res.iterations.double().mean().item():.1f}")
L.tsp2opt_debug_tune(0, 0)
