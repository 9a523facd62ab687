O=gpurun_out/r4h; mkdir -p $O
run() { tag=$1; wl=$2; extra=$3; shift; shift; shift; env "$@" python bench.py --workload $wl --steps 6 --warmup 1 --no-cpu-baseline $extra > $O/b_$tag.json 2> $O/b_$tag.err; python - <<PY
import json
d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$tag: step %.2f fact %.2f solve %.2f roofline %.2f (avg launch %.3f ms, %d launches) resid %.2e' % (d['ms_per_step'], c['factorize_ms'], c['solve_ms'], r['achieved'], r['avg_launch_ms'], r['launches'], c['solve_residual']))
PY
}
run 300k_taper 300k "" SCILMM_TUNING=1 SCILMM_DENSE_TAPER=1
run 300k_notaper 300k "" SCILMM_TUNING=1 SCILMM_DENSE_TAPER=0
run 300k_taper_ser 300k "--serialised" SCILMM_TUNING=1 SCILMM_DENSE_TAPER=1
run 300k_notaper_ser 300k "--serialised" SCILMM_TUNING=1 SCILMM_DENSE_TAPER=0
