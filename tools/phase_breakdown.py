"""Diagnostic: where does one tour's time go?  Runs a bench config with the kernel's phase stamps
enabled and prints the share of shader-clock cycles per phase (thread 0 of each workgroup).
    python tools/phase_breakdown.py [config] [dtype]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import ml4tspbench_amd as amd
from ml4tspbench_amd import _lib, two_opt

cfg = sys.argv[1] if len(sys.argv) > 1 else "tsp500"
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else np.float32
if len(sys.argv) > 4:        # override the number of instances (e.g. one workgroup per CU: no SIMD sharing)
    n_, i_, t_ = bench.CONFIGS[cfg]
    bench.CONFIGS[cfg] = (n_, int(sys.argv[4]), t_)
pts, tours, tpi = bench.make_batch(cfg, 1234, dt)
CAP = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
dev = torch.device("cuda", 0)
d_pts, d_tours = torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev)
L = _lib.lib()
amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi, max_iterations=CAP)
torch.cuda.synchronize()
L.tsp2opt_debug_phase_cycles(1)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
res = amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi, max_iterations=CAP)
t1.record(); torch.cuda.synchronize()
ws = list(two_opt._workspaces.values())[0]
st = ws[64:128].cpu().numpy().view(np.uint64)
fine = ws[128:256].cpu().numpy().view(np.uint64)
print('candidate tiles visited:', int(st[7]), 'amb events:', int(st[5]), 'exact-stage runs (single-WG) / wave0 cands (cluster)', int(st[6]), 'sweeps', int(res.sweeps.sum().item()))
names = ["sweep", "select", "apply", "stage/exchange", "tour_total"]
tot = float(st[4])
sw = int(res.sweeps.sum().item())
print(f"{cfg} {dt.__name__}: {t0.elapsed_time(t1):.2f} ms (stamped build), plan={_lib.plan(pts.shape[1], 1 if dt==np.float64 else 0, 0)}")
for n, v in zip(names, st):
    print(f"  {n:10s} {int(v):>16d} cycles  {100*float(v)/tot:6.2f} %   per sweep: {float(v)/sw:9.1f}")
if fine.any():      # TSP2OPT_LIB=.../libtsp2opt_hip_fine.so (python -m ml4tspbench_amd.build --fine)
    labels = ["A: mark+compact", "B: listed tiles", "B: closing barrier", "B: full sweep",
              "C: unit minima scan", "C: fp32 candidates", "C: decision", "C: exact stage",
              "A: reversal", "A: opening barrier", "A: closing barrier", "cluster exchange",
              "C: candidate discovery", "C: fp32 pairs of candidate tiles", "-", "-"]
    for n, v in zip(labels, fine):
        print(f"    {n:32s} per sweep: {float(v)/sw:9.1f}")
L.tsp2opt_debug_phase_cycles(0)
if pts.shape[1] > 2500 or len(tours) < 200:
    sys.exit(0)
# hot loop in isolation
K = 600
L.tsp2opt_debug_sweep_only(K)
amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi); torch.cuda.synchronize()
t0.record(); amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi); t1.record(); torch.cuda.synchronize()
L.tsp2opt_debug_sweep_only(0)
n = pts.shape[1]; pairs = (n - 1) * (n - 2) // 2
ms = t0.elapsed_time(t1)
print(f"sweep-only: {K} sweeps x {len(tours)} tours in {ms:.2f} ms -> {K*len(tours)*pairs/ms/1e6:.1f} G useful pair-deltas/s "
      f"({ms*1e-3*2.4e9*1024/(K*len(tours)*pairs/64):.1f} SIMD-cycles@2.4GHz per 64 pairs)")
