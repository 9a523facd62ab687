"""Time the selected inverse (scilmm_selected_inverse + scilmm_inverse_traces) on a BASELINE workload.
usage: python tools/sinv_timing.py 100k|300k|1m [out.json]"""
import json
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import bench
from scilmm_amd.factor import Symbolic


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "100k"
    A, C, y = bench.build_problem(name, 0)
    n = A.shape[0]
    sym = Symbolic([A, sp.identity(n, format="csr")])
    info = sym.info()
    s2 = np.array([0.4, 0.6])
    fac = sym.factorize(s2)
    t_fact = sym.timing()["factor_ms"] / 1e3
    rec = {"workload": name, "n": n, "nnzL": int(info.nnzL), "factor_flops": info.flops, "factor_s": t_fact, "runs": []}
    for rep in range(2):
        t0 = time.time()
        tr = fac.inverse_traces()
        wall = time.time() - t0
        dev = sym.timing()["quad_ms"] / 1e3
        rec["runs"].append({"wall_s": wall, "device_s": dev, "tflops": 2.0 * info.flops / dev / 1e12, "traces": tr.tolist(),
                            "trace_identity_residual": float(abs(s2 @ tr - n) / n)})
        fac.refactorize(s2)
    print(json.dumps(rec))
    if len(sys.argv) > 2:
        json.dump(rec, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
