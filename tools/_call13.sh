O=gpurun_out/r4p; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reml.py -m gpu -x -q -k "not 1m" > $O/pytest_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 $O/pytest_parity.log
if [ $rc -ne 0 ]; then tail -60 $O/pytest_parity.log; exit $rc; fi
run() { tag=$1; wl=$2; shift; shift; env "$@" python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-clean-profile > $O/b_$tag.json 2> $O/b_$tag.err; grep "virtual desc" $O/b_$tag.err | head -2; python - <<PY
import json
d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); c=d['config']
print('$tag: step %.2f fact %.2f solve %.2f update_ms %.1f resid %.1e launches %d' % (d['ms_per_step'], c['factorize_ms'], c['solve_ms'], c['update_ms'], c['solve_residual'], c['launches_per_factorize']))
PY
}
run 100k_virt 100k SCILMM_VERBOSE=1
run 100k_novirt 100k SCILMM_TUNING=1 SCILMM_VIRTUAL_MIN=100000000
run 300k_virt 300k SCILMM_VERBOSE=1
run 300k_novirt 300k SCILMM_TUNING=1 SCILMM_VIRTUAL_MIN=100000000
run 300k_virt512 300k SCILMM_TUNING=1 SCILMM_VIRTUAL_MIN=512 SCILMM_VERBOSE=1
