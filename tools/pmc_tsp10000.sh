cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_10k -- python bench.py --config tsp10000 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_10k.log 2>&1 || { tail -5 gpurun_out/pmc_10k.log; exit 1; }
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_10k/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_two_opt" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:32s} {v:.4g}")
PY
tail -1 gpurun_out/pmc_10k.log | cut -c1-300
