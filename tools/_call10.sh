O=gpurun_out/r4l; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then tail -60 $O/pytest.log; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --workload 100k --steps 20 --warmup 5 > $O/bench_100k.json 2> $O/bench_100k.err; echo "100k rc=$?"
python bench.py --workload 300k --steps 6 --warmup 1 --no-cpu-baseline > $O/bench_300k.json 2> $O/bench_300k.err; echo "300k rc=$?"
python - <<'PY'
import json
for tag in ("100k","300k"):
    d=json.loads(open("gpurun_out/r4l/bench_%s.json"%tag).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print("%s: value %.3e step %.2f fact %.2f solve %.2f roofline %.2f frac %.3f resid %.1e" % (tag, d["value"], d["ms_per_step"], c["factorize_ms"], c["solve_ms"], r["achieved"], r["frac"], c["solve_residual"]))
PY
