O=gpurun_out/r4j; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not 1m and not 100k" > $O/pytest_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 $O/pytest_parity.log
if [ $rc -ne 0 ]; then tail -40 $O/pytest_parity.log; exit $rc; fi
run() { tag=$1; wl=$2; shift; shift; env "$@" python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-clean-profile > $O/b_$tag.json 2> $O/b_$tag.err; python - <<PY
import json
d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); c=d['config']
print('$tag: step %.2f fact %.2f solve %.2f trsm %.2f potrf %.2f resid %.1e' % (d['ms_per_step'], c['factorize_ms'], c['solve_ms'], c['trsm_ms'], c['potrf_ms'], c['solve_residual']))
PY
}
run 100k_lite 100k SCILMM_TUNING=1
run 100k_old 100k SCILMM_TUNING=1 SCILMM_TRSM_LITE=0
run 300k_lite 300k SCILMM_TUNING=1
run 300k_old 300k SCILMM_TUNING=1 SCILMM_TRSM_LITE=0
SCILMM_LEVEL_DUMP=$O/levels_100k_lite.csv python bench.py --workload 100k --steps 3 --warmup 1 --no-cpu-baseline --serialised > /dev/null 2>&1
