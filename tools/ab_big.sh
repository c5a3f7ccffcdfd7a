#!/bin/bash
# tools/ab_big.sh LIB_A LIB_B -- TSP-10000 x 16 (band cluster) with two builds of the library on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms=%.3f' % (r['kernel_ms']))"; }
for lib in "$@"; do
export TSP2OPT_LIB=$PWD/ml4tspbench_amd/$lib
timeout -k 10 200 python bench.py --no-cpu --config tsp10000 --steps 2 | pr "$lib tsp10000" || exit 1
done
