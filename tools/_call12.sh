O=gpurun_out/r4o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sp in 0 1 2; do
SCILMM_TUNING=1 SCILMM_OUTSIDE_SPREAD=$sp rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$sp -o p -- python3 bench.py --workload 100k --steps 10 --warmup 2 --no-cpu-baseline --no-clean-profile > $O/bench_$sp.json 2>/dev/null; rm -f $O/prof$sp/p_kernel_trace.csv
SCILMM_TUNING=1 SCILMM_OUTSIDE_SPREAD=$sp python bench.py --workload 100k --steps 20 --warmup 3 --no-cpu-baseline --no-clean-profile > $O/b100k_$sp.json 2>/dev/null
python - <<PY
import json, pandas as pd
d=json.loads(open("$O/b100k_$sp.json").read().strip().splitlines()[-1]); c=d["config"]
k=pd.read_csv("$O/prof$sp/p_kernel_stats.csv"); k["name"]=k.Name.str.extract(r"(k_[a-z_0-9]+)")[0]
o=k[k.name=="k_outside"]
print("spread $sp: step %.2f fact %.2f resid %.1e   k_outside per factorization %.2f ms" % (d["ms_per_step"], c["factorize_ms"], c["solve_residual"], o.TotalDurationNs.sum()/1e6/13))
PY
done
