#!/bin/bash
# tools/ab_small.sh LIB_A LIB_B -- small-tour configurations with two builds of the library, interleaved on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms=%.4f' % (r['kernel_ms']))"; }
for rep in 1 2; do
for lib in "$@"; do
export TSP2OPT_LIB=$PWD/ml4tspbench_amd/$lib
timeout -k 10 120 python bench.py --no-cpu --config tsp50 --steps 10 --warmup 3 | pr "$lib tsp50" || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp100 --steps 10 --warmup 3 | pr "$lib tsp100" || exit 1
done
done
