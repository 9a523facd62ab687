O=gpurun_out/r4q; mkdir -p $O
SCILMM_VERBOSE=1 python bench.py --no-cpu-baseline --no-clean-profile --steps 3 --budget-s 220 > $O/bench_1m_virt.json 2> $O/bench_1m_virt.err
grep "virtual desc" $O/bench_1m_virt.err
SCILMM_TUNING=1 SCILMM_VIRTUAL_MIN=100000000 python bench.py --no-cpu-baseline --no-clean-profile --steps 3 --budget-s 220 > $O/bench_1m_novirt.json 2> $O/bench_1m_novirt.err
python - <<'PY'
import json
for tag in ("virt","novirt"):
    d=json.loads(open("gpurun_out/r4q/bench_1m_%s.json"%tag).read().strip().splitlines()[-1]); c=d["config"]
    print("1m %s: steps %d step %.1f fact %.1f solve %.1f update_ms %.0f launches %d resid %.2e first %.1f" % (tag, d["steps"], d["ms_per_step"], c["factorize_ms"], c["solve_ms"], c["update_ms"], c["launches_per_factorize"], c["solve_residual"], c["first_evaluation_s"]))
PY
