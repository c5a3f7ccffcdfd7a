#!/bin/bash
# tools/pmc_icache.sh -- instruction-cache counters of the headline bench (one rocprofv3 --pmc pass).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
CMD="python bench.py --steps 1 --warmup 0 --no-cpu --no-configs"
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_ic -- $CMD > gpurun_out/pmc_ic.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmc_ic.log; exit 1; }
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_two_opt" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:32s} {v:.4g}")
PY
