#!/bin/bash
# tools/quick_bench.sh -- kernel times of every BASELINE config + the small-batch TSP-500 case (no CPU leg)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'ms_per_step=%.3f kernel_ms=%.3f frac=%.3f value=%.3e' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"; }
timeout -k 10 120 python bench.py --no-cpu --no-configs --steps 5 | pr tsp500x2048 || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 16 | pr tsp500x256 || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 512 --steps 3 | pr tsp500x8192 || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp50 | pr tsp50 || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp100 | pr tsp100 || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp1000 | pr tsp1000 || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp1000 --cluster-mode 0 | pr tsp1000_1wg || exit 1
timeout -k 10 120 python bench.py --no-cpu --mode grouped | pr tsp500_grouped || exit 1
timeout -k 10 200 python bench.py --no-cpu --config tsp10000 --steps 2 | pr tsp10000 || exit 1
