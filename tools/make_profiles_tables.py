#!/usr/bin/env python3
"""tools/make_profiles_tables.py [TAG] -- the tables of profiles/README.md for round TAG (default r03), regenerated
from the committed profiles/TAG_* files (run after tools/profile_round.sh and the copy into profiles/); rewrites
the text between the BEGIN/END GENERATED TAG markers in place."""
import csv, json, os, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
PREV = {"r02": ("round 1", {"tsp50": "0.37", "tsp100": "2.47", "tsp500": "17.49", "tsp500_grouped": "17.6", "tsp1000": "16.40",
                            "tsp10000": "806.6", "pipeline_tsp50": "0.25", "pipeline_tsp100": "1.73", "pipeline_tsp500": "4.54",
                            "pipeline_sparse_tsp1000": "4.43", "pipeline_sparse_tsp10000": "61"}),
        "r03": ("round 2", {"tsp50": "0.29", "tsp100": "2.52", "tsp500": "17.32", "tsp500_grouped": "17.43", "tsp1000": "15.72",
                            "tsp10000": "709.15", "pipeline_tsp50": "0.10", "pipeline_tsp100": "0.72", "pipeline_tsp500": "1.88",
                            "pipeline_sparse_tsp1000": "1.10", "pipeline_sparse_tsp10000": "2.64"})}[TAG]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles")
J = lambda f: json.load(open(os.path.join(P, f)))
pv = PREV[1]
rows = [("tsp50", "TSP-50 x 1280", "one wave per tour", pv["tsp50"]), ("tsp100", "TSP-100 x 10000", "one wave per tour", pv["tsp100"]),
        ("tsp500", "TSP-500 x 2048", "4 waves per tour, 4 tours per CU", pv["tsp500"]),
        ("tsp500_grouped", "TSP-500 x 2048, grouped (P = 16)", "same", pv["tsp500_grouped"]),
        ("tsp1000", "TSP-1000 x 128", "2 workgroups per tour (full-window cluster)", pv["tsp1000"]),
        ("tsp10000", "TSP-10000 x 16, cap 5000", ("15 workgroups per tour (band cluster, shared rows, interleaved lane groups)" if TAG != "r02"
                                                   else "16 workgroups per tour (band cluster, shared rows)"), pv["tsp10000"])]
t1 = [f"| config (per GPU), random start tours | ms / step | pair-deltas/s | tours/s | frac of FP32 vector peak (reference-equivalent) | x 16-thread CPU port | kernel | {PREV[0]} ms |",
      "|---|---|---|---|---|---|---|---|"]
for key, name, kern, r1 in rows:
    d = J(f"{TAG}_bench_{key}.json")
    tps = d["tours_per_s"]
    tp = f"{tps / 1e6:.1f} M" if tps > 1e6 else f"{tps / 1e3:.1f} k" if tps > 1e3 else f"{tps:.1f}"
    sp = d.get("speedup_vs_cpu_baseline")
    t1.append(f"| {name} | {d['ms_per_step']:.2f} | {d['value']:.2e} | {tp} | {d['roofline']['frac']:.2f} | "
              f"{'%.0f' % sp if sp else ''} | {kern} | {r1} |")
t2 = ["| config | kernel | VALU busy | waves / SIMD | VALU wave-instructions per iteration |", "|---|---|---|---|---|"]
for key, name in (("tsp50", "TSP-50 x 1280"), ("tsp100", "TSP-100 x 10000"), ("tsp1000", "TSP-1000 x 128"), ("tsp10000", "TSP-10000 x 16")):
    d = J(f"{TAG}_{key}_pmc_valu.json")
    dv = {k.split(" ")[0]: v for k, v in d["derived"].items()}
    t2.append(f"| {name} | `{d['kernel'][0].replace('void t2o::', '')}` | {dv['valu_busy_frac']:.2f} | {dv['avg_waves_per_simd']:.1f} | "
              f"{dv['valu_wave_instructions_per_iteration']:.3g} |")
s = J(f"{TAG}_tsp500_pmc_summary.json")["derived"]
pick = lambda pre: next(v for k, v in s.items() if k.startswith(pre))
t2.insert(4, f"| TSP-500 x 2048 | `k_two_opt<double, 128, false>` | {pick('valu_busy_frac'):.2f} | {pick('avg_waves_per_simd'):.1f} | "
             f"{J(f'{TAG}_tsp500_pmc_summary.json')['counters']['SQ_INSTS_VALU'] / (J(f'{TAG}_bench_tsp500.json')['mean_iterations'] * 2048):.3g} |")
t3 = [f"| config | heat maps | decode ms | 2-opt ms | total ms | {PREV[0]} decode ms |", "|---|---|---|---|---|---|"]
for key, name, kind, r1 in (("pipeline_tsp50", "TSP-50 x 1280", "dense", pv["pipeline_tsp50"]), ("pipeline_tsp100", "TSP-100 x 10000", "dense", pv["pipeline_tsp100"]),
                            ("pipeline_tsp500", "TSP-500 x 128 x 16", "dense", pv["pipeline_tsp500"]),
                            ("pipeline_sparse_tsp1000", "TSP-1000 x 128", "k = 50 lists", pv["pipeline_sparse_tsp1000"]),
                            ("pipeline_sparse_tsp10000", "TSP-10000 x 16", "k = 50 lists", pv["pipeline_sparse_tsp10000"])):
    d = J(f"{TAG}_{key}.json")
    m = d["ms"]
    extra = ""
    if "sparse" in d:
        extra = (f" (dense decoder on the densified maps: {d['sparse']['dense_decode_ms']:.2f}; workspace "
                 f"{d['sparse']['workspace_bytes'] / 1e6:.0f} MB against {d['sparse']['dense_key_bytes'] / 1e9:.1f} GB of N x N fp64 "
                 f"keys in the reference's dense form)")
    t3.append(f"| {name} | {kind} | {m['greedy_decode']:.2f}{extra} | {m['two_opt']:.2f} | {m['total']:.2f} | {r1} |")
ks = []
for f, lab in ((f"{TAG}_kernel_stats.csv", "tsp500"), (f"{TAG}_kernel_stats_tsp100.csv", "tsp100"), (f"{TAG}_kernel_stats_tsp1000.csv", "tsp1000"),
               (f"{TAG}_kernel_stats_tsp10000.csv", "tsp10000")):
    r = list(csv.DictReader(open(os.path.join(P, f))))[0]
    ks.append(f"{lab}: `{r['Name'].split('(')[0].replace('void t2o::', '')}` {float(r['AverageNs']) / 1e6:.2f} ms x {r['Calls']} launches "
              f"({r['Percentage']} % of GPU time)")
b = J(f"{TAG}_bench_tsp500.json")
rf = b["roofline"]
head = (f"Headline (`{TAG}_bench_tsp500.json`): {b['value']:.3e} pair-deltas/s, {b['ms_per_step']:.2f} ms per step, kernel "
        f"{rf['kernel_ms']:.2f} ms (HIP events), frac {rf['frac']:.3f} reference-equivalent of the FP32 vector peak "
        f"({rf['frac_fp64_peak']:.2f} of the FP64 peak), executed_frac {rf['executed_frac']:.3f}, valu_busy {rf['valu_busy']:.2f}, "
        f"pairs_evaluated_frac {rf['pairs_evaluated_frac']:.2f}, lds_bank_conflict_rate {rf['lds_bank_conflict_rate']:.2f}, "
        f"traffic {rf['traffic'] / 1e6:.1f} MB per launch, {b['speedup_vs_cpu_baseline']:.0f}x the {b['cpu_baseline']['cores']}-thread CPU port "
        f"(GPU == oracle on its sample: {b['cpu_baseline']['gpu_matches_oracle_on_sample']}).  Library sha256 "
        f"`{J(f'{TAG}_tsp500_pmc_summary.json')['lib_sha256'][:12]}`.")
text = "\n".join([head, "", "Kernel traces: " + "; ".join(ks) + ".", "", *t1, "", *t2, "", *t3])
path = os.path.join(P, "README.md")
src = open(path).read()
new = re.sub(rf"<!-- BEGIN GENERATED {TAG} -->.*?<!-- END GENERATED {TAG} -->",
             lambda m: f"<!-- BEGIN GENERATED {TAG} -->\n" + text + f"\n<!-- END GENERATED {TAG} -->", src, flags=re.S)
assert new != src or text in src, "markers not found"
open(path, "w").write(new)
print(text)
