#!/usr/bin/env python3
"""tools/nt_probe2.py N T what=value,... [what=value,...] ... -- one random-start batch timed under several settings of
the tuning knobs (tsp2opt_debug_tune)."""
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
DEFAULT = {0: 0, 9: 1, 10: 4}
for setting in sys.argv[3:]:
    kv = dict(DEFAULT)
    for item in setting.split(","):
        if item:
            k, v = item.split("=")
            kv[int(k)] = int(v)
    for k, v in kv.items():
        L.tsp2opt_debug_tune(k, v)
    for _ in range(2):
        res = amd.two_opt_device(pts, tours, tours_per_instance=1, want_lengths=False)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        res = amd.two_opt_device(pts, tours, tours_per_instance=1, want_lengths=False)
    ev[1].record(); torch.cuda.synchronize()
    print(f"N={n} T={T} [{setting}]: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms")
for k, v in DEFAULT.items():
    L.tsp2opt_debug_tune(k, v)
