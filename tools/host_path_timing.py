"""Per-evaluation time of the reference-shaped Python path (host buffers in and out, PCIe included) next to the
device-resident step `bench.py` times.  usage: python tools/host_path_timing.py <workload: 100k|300k>"""
import importlib
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

M = importlib.import_module("scilmm_amd.SparseCholesky")
from scilmm_amd.harness.pedigree import make_problem  # noqa: E402


def main():
    name = sys.argv[1]
    n, sf = bench.WORKLOADS[name]
    mats, C, y = make_problem(n, sf, seed=0)
    import scipy.sparse as sp
    mats = mats + [sp.identity(y.size, format="csr")]
    chol = M.SparseCholesky()
    x = np.log(np.array([0.4, 0.6]))
    np.random.seed(0)
    t0 = time.time()
    M.bolt_gradient_estimation(x, chol, mats, C, y, True, 100, False)
    first = time.time() - t0
    ts = []
    for k in range(4):
        t0 = time.time()
        nll, g = M.bolt_gradient_estimation(x + 0.01 * k, chol, mats, C, y, True, 100, False)
        ts.append(time.time() - t0)
    sym = chol.engine_for(mats)
    print(json.dumps({"workload": name, "n": int(y.size), "nnzL": int(sym.info().nnzL), "first_evaluation_s": first,
                      "evaluation_s": float(np.median(ts)), "nnzL_per_s_host_path": sym.info().nnzL / float(np.median(ts)),
                      "note": "bolt_gradient_estimation through host buffers: randn(n,100) on the host (overlapped with the "
                              "factorization), L*R, one 103-column solve, K fused SpMM+reduce calls, all PCIe transfers"}))


if __name__ == "__main__":
    main()
