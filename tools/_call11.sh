O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not 1m" > $O/pytest_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 $O/pytest_parity.log
if [ $rc -ne 0 ]; then tail -40 $O/pytest_parity.log; exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof100k -o p -- python3 bench.py --workload 100k --steps 20 --warmup 2 --no-cpu-baseline --no-clean-profile > $O/bench_100k.json 2>/dev/null; rm -f $O/prof100k/p_kernel_trace.csv
python bench.py --workload 100k --steps 20 --warmup 3 --no-cpu-baseline --no-clean-profile > $O/b100k.json 2>/dev/null
python bench.py --workload 300k --steps 6 --warmup 1 --no-cpu-baseline --no-clean-profile > $O/b300k.json 2>/dev/null
python - <<'PY'
import json, pandas as pd
for tag in ("b100k","b300k"):
    d=json.loads(open("gpurun_out/r4n/%s.json"%tag).read().strip().splitlines()[-1]); c=d["config"]
    print("%s: step %.2f fact %.2f solve %.2f resid %.1e" % (tag, d["ms_per_step"], c["factorize_ms"], c["solve_ms"], c["solve_residual"]))
k=pd.read_csv("gpurun_out/r4n/prof100k/p_kernel_stats.csv")
k["name"]=k.Name.str.extract(r"(k_[a-z_0-9]+)")[0]
print(k[k.name.isin(["k_outside","k_trsm_lite","k_potrf","k_dense_b"])][["name","Calls","TotalDurationNs","AverageNs"]].to_string())
PY
