"""Summarise gpurun_out/pmc_*/ counter CSVs for the persistent kernel into one JSON/markdown."""
import csv, glob, json, sys, collections
agg = collections.defaultdict(float); ndisp = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_two_opt" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if r.get("End_Timestamp") else 0
out = dict(agg)
simd_cycles = out.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024      # GRBM summed over 8 XCDs; 1024 SIMDs
d = {
  "counters": out,
  "derived": {
    "valu_insts_per_simd_cycle": out.get("SQ_INSTS_VALU", 0) / simd_cycles if simd_cycles else None,
    "valu_busy_frac (ACTIVE_INST_VALU*4 / SIMD cycles)": out.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles if simd_cycles else None,
    "lds_bank_conflict_rate (BANK_CONFLICT / LDS_IDX_ACTIVE)": out.get("SQ_LDS_BANK_CONFLICT", 0) / out["SQ_LDS_IDX_ACTIVE"] if out.get("SQ_LDS_IDX_ACTIVE") else None,
    "avg_waves_per_simd (WAVE_CYCLES*4 / SIMD cycles)": out.get("SQ_WAVE_CYCLES", 0) * 4 / simd_cycles if simd_cycles else None,
    "fp32_flop_executed (FMA*2+ADD+MUL+TRANS, x64 lanes)": 64 * (2 * out.get("SQ_INSTS_VALU_FMA_F32", 0) + out.get("SQ_INSTS_VALU_ADD_F32", 0) + out.get("SQ_INSTS_VALU_MUL_F32", 0) + out.get("SQ_INSTS_VALU_TRANS_F32", 0)),
    "hbm_bytes (FETCH_SIZE*2 + WRITE_SIZE) KiB->B, gfx950 read correction": (2 * out.get("FETCH_SIZE", 0) + out.get("WRITE_SIZE", 0)) * 1024,
  },
}
print(json.dumps(d, indent=1))
