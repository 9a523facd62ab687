O=gpurun_out/r4f; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --workload 100k --steps 20 --warmup 3 --no-cpu-baseline --no-clean-profile > $O/b_$tag.json 2> $O/b_$tag.err; python - <<PY
import json
d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); c=d['config']
print('$tag: step %.2f fact %.2f solve %.2f' % (d['ms_per_step'], c['factorize_ms'], c['solve_ms']))
PY
}
run base SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=1
run c16_hi SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=16
run c16_lo SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=16 SCILMM_OUTSIDE_PRIO=0
run c4_lo SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=4 SCILMM_OUTSIDE_PRIO=0
run c64_lo SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=64 SCILMM_OUTSIDE_PRIO=0
run c1_lo SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=1 SCILMM_OUTSIDE_PRIO=0
SCILMM_TUNING=1 SCILMM_OUTSIDE_CHUNKS=16 SCILMM_OUTSIDE_PRIO=0 SCILMM_LEVEL_DUMP=$O/levels_c16_lo.csv python bench.py --workload 100k --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
