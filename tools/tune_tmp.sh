cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in tsp100 tsp500; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gk_$c -- python tools/pipeline_bench.py --config $c --check 0 > /dev/null 2>&1
python - $(find gpurun_out/gk_$c -name "*kernel_stats.csv" | head -1) $c <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(sys.argv[2], r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e6,3), "ms")
PY
done
