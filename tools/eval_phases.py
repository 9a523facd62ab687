"""Where one REML evaluation's wall-clock goes beyond factorize + solve: runs the shared evaluation a few times on a BASELINE
workload with per-evaluation metrics on and prints wall seconds next to the device timers.
usage: python tools/eval_phases.py 100k|300k|1m [evaluations] [out.jsonl]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "100k"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/eval_phases.jsonl"
    if os.path.exists(out):
        os.remove(out)
    import importlib
    P = importlib.import_module("scilmm_amd.SparseCholesky")  # (the package re-exports the class under the same name)
    import scipy.sparse as sp
    A, C, y = bench.build_problem(name, 0)
    y = y / y.std()
    mats = [A, sp.eye(A.shape[0]).tocsr()]
    chol = P.SparseCholesky(metrics=out)
    np.random.seed(1)
    for i in range(reps):
        t0 = time.time()
        nll, g = P.bolt_gradient_estimation(np.log([0.4 + 0.01 * i, 0.6]), chol, mats, C, y, True, 100, False)
        print("evaluation %d: %.3f s wall, nll %.6f" % (i, time.time() - t0, nll), flush=True)
    for line in open(out):
        r = json.loads(line)
        d = r["device_ms"]
        print(json.dumps({"evaluation": r["evaluation"], "seconds": round(r["seconds"], 3),
                          "device_ms": {k: round(v, 1) for k, v in d.items()},
                          "device_sum_s": round(sum(d.values()) / 1e3, 3)}))


if __name__ == "__main__":
    main()
