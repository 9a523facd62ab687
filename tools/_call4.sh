set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SCILMM_VERBOSE=1 python bench.py --workload 100k --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_100k.json 2> $O/bench_100k.err; rc=$?; echo "bench100k rc=$rc"; grep "k_outside" $O/bench_100k.err | head -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4e/bench_100k.json').read().strip().splitlines()[-1]); c=d['config']
print('100k: step', d['ms_per_step'], 'fact', c['factorize_ms'], 'solve', c['solve_ms'], 'resid', c['solve_residual'])
PY
python bench.py --workload 300k --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_300k.json 2> $O/bench_300k.err; rc=$?; echo "bench300k rc=$rc"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4e/bench_300k.json').read().strip().splitlines()[-1]); c=d['config']
print('300k: step', d['ms_per_step'], 'fact', c['factorize_ms'], 'solve', c['solve_ms'], 'resid', c['solve_residual'])
PY
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
