#!/usr/bin/env python3
"""tools/summarize_pmc_decoder.py sq.csv fetch.csv write.csv -- three rocprofv3 --pmc passes of
`tools/pipeline_bench.py --config tsp500 --check 0 --steps 1` reduced to per-launch figures of the dense
decoder's kernels: VALU busy, waves per SIMD, HBM bytes (FETCH_SIZE*2 + WRITE_SIZE, KiB -> B), and for the
prepare kernel the VALU wave-instructions per 64 key pairs."""
import collections, csv, hashlib, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"command": "three rocprofv3 --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE) of python tools/pipeline_bench.py "
                  "--config tsp500 --check 0 --steps 1 (2048 dense heat maps of 500 nodes, fp32; two decoder launches "
                  "per pass, figures per launch)",
       "lib_sha256": hashlib.sha256(open(os.path.join(root, "ml4tspbench_amd", "libtsp2opt_hip.so"), "rb").read()).hexdigest(),
       "kernels": {}}
for path in sys.argv[1:4]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void t2o::", "")
        if "greedy" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
                cnt[k] += 1
    for k, v in acc.items():
        e = out["kernels"].setdefault(k, {})
        for kk, vv in v.items():
            e[kk + "_per_launch"] = vv / cnt[k]
for k, e in out["kernels"].items():
    g = e["GRBM_GUI_ACTIVE_per_launch"]
    e["valu_busy_frac"] = e["SQ_ACTIVE_INST_VALU_per_launch"] * 4 / (g / 8 * 1024)
    e["avg_waves_per_simd"] = e["SQ_WAVE_CYCLES_per_launch"] * 4 / (g / 8 * 1024)
    e["hbm_bytes_per_launch (FETCH_SIZE*2 + WRITE_SIZE, KiB -> B)"] = (2 * e["FETCH_SIZE_per_launch"] + e["WRITE_SIZE_per_launch"]) * 1024
    if "prepare" in k:
        e["valu_wave_instructions_per_64_pairs"] = e["SQ_INSTS_VALU_per_launch"] / (2048 * 500 * 500 / 64)
print(json.dumps(out, indent=1))
