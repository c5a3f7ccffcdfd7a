#!/bin/bash
# tools/pmc_trans.sh [config] -- one rocprofv3 --pmc pass over a bench configuration (default tsp10000) with the
# square-root counter: SQ_INSTS_VALU_TRANS_F32 x 64 = pairs the filter really evaluated (one v_sqrt_f32 per pair)
export CFG=${1:-tsp10000}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$CFG -- python bench.py --config $CFG --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_$CFG.log 2>&1 || { tail -5 gpurun_out/pmc_$CFG.log; exit 1; }
python - <<'PY'
import csv, glob, collections, json, os
cfg = os.environ["CFG"]
tot = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/pmc_{cfg}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_two_opt" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{cfg} {k:32s} {v:.4g}")
line = [l for l in open(f"gpurun_out/pmc_{cfg}.log") if l.startswith("{")]
if line:
    d = json.loads(line[-1])
    pd = d["value"] * d["ms_per_step"] * 1e-3          # reference pair-deltas of the one launch
    ev = tot["SQ_INSTS_VALU_TRANS_F32"] * 64
    print(f"{cfg} reference pair-deltas {pd:.4g}, evaluated (sqrt x 64) {ev:.4g}: fraction {ev / pd:.3f}")
PY
