#!/usr/bin/env python3
"""tools/late_phase_probe.py -- latency of LATE 2-opt iterations (short reversals): greedy-decoded TSP-500 start tours,
one workgroup per CU (256 tours) against four per CU (2048 tours); prints microseconds per iteration of the slowest slot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ml4tspbench_amd as amd
dev = torch.device("cuda", 0)
for inst in (16, 128):
    rng = np.random.default_rng(1234)
    pts = torch.from_numpy(rng.random((inst, 500, 2))).to(dev)
    heat = bench.make_heat(pts, 16, 0, 1234)
    tours = amd.greedy_decode_device(heat, pts, maps_per_instance=16)
    for _ in range(3):
        res = amd.two_opt_device(pts, tours, tours_per_instance=16)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        res = amd.two_opt_device(pts, tours, tours_per_instance=16, want_lengths=False)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    it = res.iterations.double()
    rounds = max(1.0, inst * 16 / 1024)
    print(f"{inst*16} tours: {ms:.3f} ms per call, iterations mean {it.mean().item():.1f} max {int(it.max().item())}; "
          f"{ms*1e3/(it.mean().item()*rounds):.2f} us per iteration and resident workgroup")

if "fine" in os.environ.get("TSP2OPT_LIB", ""):   # (fineB / fineC / fineM: only that phase's marks compiled in)       # diagnostic build: phase stamps of the 256-tour case
    from ml4tspbench_amd import _lib, two_opt
    L = _lib.lib()
    rng = np.random.default_rng(1234)
    pts = torch.from_numpy(rng.random((16, 500, 2))).to(dev)
    heat = bench.make_heat(pts, 16, 0, 1234)
    tours = amd.greedy_decode_device(heat, pts, maps_per_instance=16)
    L.tsp2opt_debug_phase_cycles(1)
    res = amd.two_opt_device(pts, tours, tours_per_instance=16)
    torch.cuda.synchronize()
    ws = list(two_opt._workspaces.values())[0]
    st = ws[64:128].cpu().numpy().view(np.uint64)
    fine = ws[128:256].cpu().numpy().view(np.uint64)
    sw = int(res.sweeps.sum().item())
    print("sweeps", sw, "per sweep: sweep %.0f select %.0f apply %.0f stage %.0f total %.0f" % tuple(float(v) / sw for v in st[:5]))
    labels = ["A: mark+compact", "B: listed tiles", "B: closing barrier", "B: full sweep", "C: unit minima scan",
              "C: fp32 candidates", "C: decision", "C: exact stage", "A: reversal", "A: opening barrier", "A: closing barrier"]
    for n_, v in zip(labels, fine):
        print(f"    {n_:24s} {float(v)/sw:9.1f}")
    L.tsp2opt_debug_phase_cycles(0)
