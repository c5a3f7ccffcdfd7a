#!/bin/bash
# tools/xw_scan.sh -- unit-less waves of the wide geometry (tuning knob 10) at TSP-500 x 256
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms=%.3f' % (r['kernel_ms']))"; }
for x in 2 4 6 8; do
timeout -k 10 120 python bench.py --no-cpu --instances 16 --tune 10=$x | pr "tsp500x256 extra=$x" || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 8 --tune 10=$x | pr "tsp500x128 extra=$x" || exit 1
done
