set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_reml.py tests/test_aireml.py -m gpu -x -q > $O/pytest_reml.log 2>&1; rc=$?; echo "reml tests rc=$rc"; tail -15 $O/pytest_reml.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/fit_timing.py 100k --out $O/fit_100k.json --compare profiles/r3_fit_100k.json > $O/fit_100k.log 2>&1; rc=$?; echo "fit100k rc=$rc"; tail -4 $O/fit_100k.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 python tools/fit_timing.py 1m --out $O/fit_1m.json --compare profiles/r3_fit_1m.json > $O/fit_1m.log 2>&1; rc=$?; echo "fit1m rc=$rc"; tail -4 $O/fit_1m.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serialised -o ser -- python3 bench.py --workload 1m --serialised --steps 2 --warmup 0 --no-cpu-baseline > $O/bench_1m_serialised.json 2> $O/bench_1m_serialised.err; rc=$?; echo "serialised rc=$rc"; tail -c 300 $O/bench_1m_serialised.err
ls $O/serialised | head; rm -f $O/serialised/ser_kernel_trace.csv
