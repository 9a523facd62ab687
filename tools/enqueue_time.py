"""How long the HOST needs to queue one factorization (scilmm_refactorize_async returns when everything is enqueued) against
how long the device needs to run it -- with and without the engine's HIP-event brackets.  usage: enqueue_time.py 100k|300k"""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scilmm_amd.factor import Symbolic
name = sys.argv[1] if len(sys.argv) > 1 else "100k"
A, C, y = bench.build_problem(name, 0)
n = A.shape[0]
sym = Symbolic([A, sp.identity(n, format="csr")])
fac = sym.factorize([0.4, 0.6])
for prof in (0, 1):
    sym.set_profiling(prof)
    fac.refactorize([0.4, 0.6])
    enq, tot = [], []
    for i in range(10):
        t0 = time.perf_counter()
        fac.refactorize_async([0.4 + 0.001 * i, 0.6])
        t1 = time.perf_counter()
        fac.wait()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    tm = sym.timing()
    print("%s profiling=%d: host enqueue %.2f ms (min %.2f), whole %.2f ms, device factor_ms %.2f, launches %d" %
          (name, prof, 1e3 * np.median(enq), 1e3 * min(enq), 1e3 * np.median(tot), tm["factor_ms"], tm["n_launches"]))
