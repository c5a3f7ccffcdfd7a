#!/bin/bash
# tools/lcap_scan.sh -- changed-tile list capacity (tuning knob 4, sixteenths of the tiles) at the headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms=%.3f' % (r['kernel_ms']))"; }
for x in 13 10 11 12 14 15 16 13; do
timeout -k 10 120 python bench.py --no-cpu --no-configs --steps 5 --tune 4=$x | pr "tsp500x2048 lcap16s=$x" || exit 1
done
