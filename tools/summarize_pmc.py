"""Summarise gpurun_out/pmc_*/ counter CSVs for the persistent kernel into one JSON.

Run right after tools/pmc_passes.sh on the same box, from the repo root.  The summary carries the
sha256 of ml4tspbench_amd/libtsp2opt_hip.so: bench.py only quotes these counters for the library they
were measured on (and prints null for a different build)."""
import csv, glob, hashlib, json, os, collections
agg = collections.defaultdict(float)
launches = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_two_opt" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            launches[f].add(r.get("Dispatch_Id", "0"))
out = dict(agg)
nl = sorted({len(v) for v in launches.values()})
simd_cycles = out.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024      # GRBM summed over 8 XCDs; 1024 SIMDs
lib = os.path.join("ml4tspbench_amd", "libtsp2opt_hip.so")
d = {
  "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
  "command": "python bench.py --steps 1 --warmup 0 --no-cpu --no-configs (one rocprofv3 --pmc run per pass, tools/pmc_passes.sh)",
  "k_two_opt_launches_per_pass": nl,
  "counters": out,
  "derived": {
    "valu_insts_per_simd_cycle": out.get("SQ_INSTS_VALU", 0) / simd_cycles if simd_cycles else None,
    "valu_busy_frac (ACTIVE_INST_VALU*4 / SIMD cycles)": out.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles if simd_cycles else None,
    "lds_bank_conflict_rate (BANK_CONFLICT / LDS_IDX_ACTIVE)": out.get("SQ_LDS_BANK_CONFLICT", 0) / out["SQ_LDS_IDX_ACTIVE"] if out.get("SQ_LDS_IDX_ACTIVE") else None,
    "avg_waves_per_simd (WAVE_CYCLES*4 / SIMD cycles)": out.get("SQ_WAVE_CYCLES", 0) * 4 / simd_cycles if simd_cycles else None,
    "fp32_flop_executed (FMA*2+ADD+MUL+TRANS, x64 lanes)": 64 * (2 * out.get("SQ_INSTS_VALU_FMA_F32", 0) + out.get("SQ_INSTS_VALU_ADD_F32", 0) + out.get("SQ_INSTS_VALU_MUL_F32", 0) + out.get("SQ_INSTS_VALU_TRANS_F32", 0)),
    "sqrt_lanes (TRANS_F32 x64 = pair-deltas actually evaluated by the filter)": 64 * out.get("SQ_INSTS_VALU_TRANS_F32", 0),
    "hbm_bytes (FETCH_SIZE*2 + WRITE_SIZE) KiB->B, gfx950 read correction": (2 * out.get("FETCH_SIZE", 0) + out.get("WRITE_SIZE", 0)) * 1024,
  },
  "note": "per launch (one k_two_opt launch per pass); a packed v_pk_* instruction is one count in the "
          "SQ_INSTS_VALU_*_F32 counters although it produces two results, so fp32_flop_executed is a lower bound",
}
print(json.dumps(d, indent=1))
