"""Diagnostic: where does the greedy decoder's merge loop spend its time?
    python tools/greedy_breakdown.py [config] [knn]
Prints per-phase cycle shares of k_greedy's rounds (tsp2opt_debug_greedy_cycles) for a pipeline_bench config.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import pipeline_bench as pb
import ml4tspbench_amd as amd
from ml4tspbench_amd import _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "tsp500"
knn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, inst, per = pb.CONFIGS[cfg]
dev = torch.device("cuda", 0)
pts = torch.from_numpy(np.random.default_rng(1234).random((inst, n, 2))).to(dev)
heat = pb.make_heat(pts, per, knn, 1234)
amd.greedy_decode_device(heat, pts, maps_per_instance=per)
torch.cuda.synchronize()
L = _lib.lib()
L.tsp2opt_debug_greedy_cycles.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
if os.environ.get("GREEDY_K1") is not None:
    L.tsp2opt_debug_greedy_cycles(2 if os.environ["GREEDY_K1"] == "1" else 3, 0, None)
    amd.greedy_decode_device(heat, pts, maps_per_instance=per)
L.tsp2opt_debug_greedy_cycles(1, 0, None)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); amd.greedy_decode_device(heat, pts, maps_per_instance=per); t1.record()
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 8)()
L.tsp2opt_debug_greedy_cycles(0, 0, out)
adv, resc, apply_, nm, nr, tot, rounds, serial = [int(out[k]) for k in range(8)]
maps = inst * per
print(f"{cfg}: {t0.elapsed_time(t1):.2f} ms, {maps} maps, merges/map {nm/maps:.1f}, rescans/map {nr/maps:.1f}, "
      f"rounds/map {rounds/maps:.1f} ({serial/maps:.2f} with one exactly compared edge)")
print(f"  cycles per map: total {tot/maps:.0f}; list advances {adv/maps:.0f}; rescans {resc/maps:.0f} "
      f"({resc/max(nr,1):.0f}/rescan); choose+apply {apply_/maps:.0f} ({apply_/max(rounds,1):.0f}/round)")
