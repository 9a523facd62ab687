O=gpurun_out/r4s; mkdir -p $O
run() { tag=$1; shift; env "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python - <<PY
import json
d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$tag: step %.1f fact %.1f solve %.1f trsm %.1f potrf %.1f refine [%s] resid %.1e roofline %.1f' % (d['ms_per_step'], c['factorize_ms'], c['solve_ms'], c['trsm_ms'], c['potrf_ms'], c['refinement'][:60], c['solve_residual'], (r or {}).get('achieved',0)))
PY
}
run 300k_k3_fp32_lite python bench.py --workload 300k --components A,D --front-bits 32 --steps 5 --warmup 1 --no-cpu-baseline
run 300k_k3_fp32_old SCILMM_TUNING=1 SCILMM_TRSM_LITE=0 python bench.py --workload 300k --components A,D --front-bits 32 --steps 5 --warmup 1 --no-cpu-baseline
run 1m_fp32_lite python bench.py --front-bits 32 --steps 4 --no-cpu-baseline --budget-s 230
run 1m_fp32_old SCILMM_TUNING=1 SCILMM_TRSM_LITE=0 python bench.py --front-bits 32 --steps 4 --no-cpu-baseline --budget-s 230
