#!/usr/bin/env python3
"""tools/summarize_pmc_valu.py CONFIG counter_collection.csv bench_line.json TOURS -- one SQ counter pass of
`bench.py --config CONFIG --steps 1 --warmup 0 --no-cpu` reduced to VALU busy, waves per SIMD and VALU / LDS
wave-instructions per 2-opt iteration (iterations = mean_iterations x tours of the bench line)."""
import collections, csv, hashlib, json, os, sys
cfg, cc, bench, tours = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(float)
names = set()
for r in csv.DictReader(open(cc)):
    if "k_two_opt" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        names.add(r["Kernel_Name"].split("(")[0])
g = acc["GRBM_GUI_ACTIVE"]
b = json.loads([l for l in open(bench).read().splitlines() if l.startswith("{")][-1])
iters = b["mean_iterations"] * tours
print(json.dumps({
    "command": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS "
               f"SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE -- python bench.py --config {cfg} --steps 1 --warmup 0 "
               f"--no-cpu (one launch)",
    "kernel": sorted(names),
    "lib_sha256": hashlib.sha256(open(os.path.join(root, "ml4tspbench_amd", "libtsp2opt_hip.so"), "rb").read()).hexdigest(),
    "counters": dict(acc),
    "derived": {"valu_busy_frac (ACTIVE_INST_VALU*4 / SIMD cycles of the whole chip)": acc["SQ_ACTIVE_INST_VALU"] * 4 / (g / 8 * 1024),
                "avg_waves_per_simd": acc["SQ_WAVE_CYCLES"] * 4 / (g / 8 * 1024),
                "iterations": iters,
                "valu_wave_instructions_per_iteration": acc["SQ_INSTS_VALU"] / iters,
                "lds_wave_instructions_per_iteration": acc["SQ_INSTS_LDS"] / iters}}, indent=1))
