"""The N-GPU bench gives rank r the pedigree of seed r: check a few seeds end to end on one GPU."""
import sys, ctypes, time
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
import bench
from scilmm_amd.factor import Symbolic
for seed in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    A, C, y = bench.build_problem("100k", seed)
    n = A.shape[0]
    sym = Symbolic([A, sp.identity(n, format="csr")])
    info = sym.info()
    f = sym.factorize([0.4, 0.6])
    B = np.random.default_rng(seed).standard_normal((n, 103))
    t = time.time(); X = f(B); dt = time.time() - t
    r = np.abs(0.4 * (A @ X) + 0.6 * X - B).max() / np.abs(B).max()
    f.refactorize([0.41, 0.59]); 
    tm = sym.timing()
    print("seed %d: n=%d nnzL=%.3e flops=%.3e levels=%d  factorize %.1f ms  solve residual %.2e  logdet %.6f" % (
        seed, n, info.nnzL, info.flops, info.nlevels, tm["factor_ms"], r, f.logdet()), flush=True)
    del f, sym
