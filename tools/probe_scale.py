"""Symbolic-only probe of a bench workload: sizes of the factor and of the plan inputs, host memory high-water mark.
usage: python tools/probe_scale.py 1m"""
import resource, sys, time
sys.path.insert(0, ".")
import scipy.sparse as sp
import bench
from scilmm_amd.factor import Symbolic
name = sys.argv[1] if len(sys.argv) > 1 else "300k"
t = time.time()
A, C, y = bench.build_problem(name, 0)
n = A.shape[0]
print("problem %s: n=%d nnz(A)=%.3e  %.1f s  maxrss %.1f GB" % (name, n, A.nnz, time.time() - t, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
t = time.time()
sym = Symbolic([A, sp.identity(n, format="csr")], upload=False)
i = sym.info()
print("symbolic %.1f s: nnzL=%.3e stored=%.3e (%.1f GB) flops=%.3e nsuper=%d levels=%d  maxrss %.1f GB" % (
    time.time() - t, i.nnzL, i.nnzL_stored, i.nnzL_stored * 8 / 1e9, i.flops, i.nsuper, i.nlevels,
    resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
print("combos %.3e  update pairs %.3e" % (len(sym.get("combo_pair")), len(sym.get("upd_src"))), flush=True)
