#!/usr/bin/env python3
"""tests/fuzz/scan_cluster_sizes.py -- every n in a range through the forced cluster kernels (2, 4, 8 workgroups
per tour) against the CPU oracle; prints the sizes (with their planned geometry) whose tours or iteration
counts differ.  Found nothing after the planner fix of round 2 (n = 40 .. 700)."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ml4tspbench_amd as amd
from ml4tspbench_amd import _lib
from oracle import two_opt_oracle
L=_lib.lib(); dev=torch.device("cuda",0)
bad={}
for mode in (2,4,8):
    L.tsp2opt_debug_cluster_mode(mode)
    for n in list(range(40,420,1))+list(range(420,700,7)):
        out=(ctypes.c_int32*12)()
        L.tsp2opt_cluster_plan(n, 1, 4, 256, out)
        nb=0
        for seed in range(2):
            rng=np.random.default_rng(seed*1000+n)
            pts=rng.random((1,n,2)); T=4
            tours=np.zeros((T,n+1),dtype=np.int64)
            for k in range(T): tours[k,1:n]=rng.permutation(np.arange(1,n))
            res=amd.two_opt_device(torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev), tours_per_instance=4, max_iterations=5000, want_lengths=False)
            rt,ri,rs=two_opt_oracle.solve(pts,tours,tours_per_instance=4,max_iterations=5000,threads=8)
            gi=res.iterations.cpu().numpy()
            if not (np.array_equal(res.tours.cpu().numpy(),rt) and np.array_equal(gi,ri)): nb+=1
        if nb: bad[(mode,n)]=(nb,list(out))
L.tsp2opt_debug_cluster_mode(-1)
for k,v in bad.items(): print(k,v)
print("bad count",len(bad))
