set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_zz_gpu_dist.py -m gpu -x -q > $O/pytest_dist.log 2>&1; rc=$?; echo "dist tests rc=$rc"; tail -3 $O/pytest_dist.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 900 python bench.py --plan-only --gpus 8 --workload 3m --components A,D --front-bits 32 > $O/plan_3m_w8_k3_fp32.json 2> $O/plan_3m.err; rc=$?; echo "plan3m rc=$rc"; tail -c 600 $O/plan_3m.err; head -c 300 $O/plan_3m_w8_k3_fp32.json
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_dense_b" --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --workload 1m --steps 1 --warmup 0 --no-cpu-baseline --no-engine-profiling --dump-maps $O/maps_fetch.txt > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; rc=$?; echo "pmc fetch rc=$rc"; tail -c 1500 $O/pmc_fetch.err
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "k_dense_b" --output-format csv -d $O/pmc_write -o write -- python3 bench.py --workload 1m --steps 1 --warmup 0 --no-cpu-baseline --no-engine-profiling > $O/pmc_write_bench.json 2> $O/pmc_write.err; rc=$?; echo "pmc write rc=$rc"; tail -c 600 $O/pmc_write.err
ls -la $O/pmc_fetch $O/pmc_write 2>/dev/null | head -20
