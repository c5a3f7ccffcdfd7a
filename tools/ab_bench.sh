#!/bin/bash
# tools/ab_bench.sh LIB_A LIB_B -- the same kernel timings with two builds of the library, interleaved on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms=%.3f' % (r['kernel_ms']))"; }
for rep in 1 2; do
for lib in "$@"; do
export TSP2OPT_LIB=$PWD/ml4tspbench_amd/$lib
timeout -k 10 120 python bench.py --no-cpu --no-configs --steps 5 | pr "$lib tsp500x2048" || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 512 --steps 3 | pr "$lib tsp500x8192" || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 16 | pr "$lib tsp500x256" || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp100 | pr "$lib tsp100" || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp1000 | pr "$lib tsp1000" || exit 1
done
done
