cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms=%.3f' % (r['kernel_ms']))"; }
for x in 0 4 6 8; do
timeout -k 10 120 python bench.py --no-cpu --instances 16 --tune 10=$x | pr "tsp500x256 extra=$x" || exit 1
done
python - <<'PY'
import subprocess, sys
PY
