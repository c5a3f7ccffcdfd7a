cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'ms_per_step=%.3f kernel_ms=%.3f' % (d['ms_per_step'], r['kernel_ms']))"; }
timeout -k 10 120 python bench.py --no-cpu --no-configs --steps 5 | pr tsp500x2048 || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 512 --steps 3 | pr tsp500x8192 || exit 1
for nt in 256 512 1024; do
timeout -k 10 120 python bench.py --no-cpu --instances 16 --tune 0=$nt | pr tsp500x256_nt$nt || exit 1
timeout -k 10 120 python bench.py --no-cpu --instances 32 --tune 0=$nt | pr tsp500x512_nt$nt || exit 1
done
timeout -k 10 120 python bench.py --no-cpu --no-configs --steps 5 --tune 0=512 | pr tsp500x2048_nt512 || exit 1
timeout -k 10 120 python bench.py --no-cpu --config tsp1000 --cluster-mode 0 --tune 0=1024 | pr tsp1000_1wg_nt1024 || exit 1
timeout -k 10 120 python tools/late_phase_probe.py
