O=gpurun_out/r4k; mkdir -p $O
SCILMM_TUNING=1 SCILMM_TRSM_LITE=0 python bench.py --no-cpu-baseline --no-clean-profile --steps 3 --budget-s 200 > $O/bench_1m_oldtrsm.json 2> $O/bench_1m_oldtrsm.err
python bench.py --no-cpu-baseline --no-clean-profile --steps 3 --budget-s 200 > $O/bench_1m_lite.json 2> $O/bench_1m_lite.err
python - <<'PY'
import json
for tag in ("oldtrsm","lite"):
    d=json.loads(open("gpurun_out/r4k/bench_1m_%s.json"%tag).read().strip().splitlines()[-1]); c=d["config"]
    print("1m %s: steps %d step %.1f fact %.1f solve %.1f trsm %.1f potrf %.1f resid %.2e" % (tag, d["steps"], d["ms_per_step"], c["factorize_ms"], c["solve_ms"], c["trsm_ms"], c["potrf_ms"], c["solve_residual"]))
PY
