O=gpurun_out/r4g; mkdir -p $O
run() { tag=$1; wl=$2; shift; shift; env "$@" python bench.py --workload $wl --steps 6 --warmup 1 --no-cpu-baseline --no-clean-profile > $O/b_$tag.json 2> $O/b_$tag.err; python - <<PY
import json
d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); c=d['config']
print('$tag: step %.2f fact %.2f solve %.2f' % (d['ms_per_step'], c['factorize_ms'], c['solve_ms']))
PY
}
run 300k_base 300k SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=1
run 300k_c4_lo 300k SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=4 SCILMM_OUTSIDE_PRIO=0
run 300k_c16_lo 300k SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=16 SCILMM_OUTSIDE_PRIO=0
run 300k_c16_hi 300k SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=16
run 300k_c8_lo 300k SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=8 SCILMM_OUTSIDE_PRIO=0
