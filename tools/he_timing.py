"""Wall-clock of the Haseman-Elston estimator WITH its Monte-Carlo standard error (reference SparseCholesky.py:192-281, what
`run_estimates(reml=False)` runs) with the n x 100 products on the device (scilmm_csr_spmm_dev) and -- optionally -- on the host
(SCILMM_HOST_BUFFERS=1: SciPy), same np.random stream.  usage: he_timing.py 100k|300k|1m [--host] [--out FILE]"""
import argparse, importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser()
ap.add_argument("workload"); ap.add_argument("--host", action="store_true"); ap.add_argument("--out", default=None)
args = ap.parse_args()
P = importlib.import_module("scilmm_amd.SparseCholesky")
A, C, y = bench.build_problem(args.workload, 0)
rec = {"workload": args.workload, "n": int(A.shape[0]), "nnz_A": int(A.nnz), "sim_num": 100}
for host in ([False, True] if args.host else [False]):
    if host:
        os.environ["SCILMM_HOST_BUFFERS"] = "1"
    else:
        os.environ.pop("SCILMM_HOST_BUFFERS", None)
    np.random.seed(5)
    t0 = time.time()
    est, std = P.HE([A], C, y, compute_stderr=True)
    dt = time.time() - t0
    rec["host" if host else "device"] = {"seconds": dt, "he": est.tolist(), "std": std.tolist()}
    print("%s products: %.2f s, h2 %.6f +- %.6f" % ("host SciPy" if host else "device", dt, est[0], std[0]), flush=True)
if args.host:
    rec["std_rel_diff"] = abs(rec["host"]["std"][0] - rec["device"]["std"][0]) / abs(rec["host"]["std"][0])
print(json.dumps(rec))
if args.out:
    json.dump(rec, open(args.out, "w"), indent=1)
