"""End-to-end REML fit timing on a BASELINE config through the drop-in Python surface (not the bench metric).
usage: python tools/fit_timing.py [100k|10k]"""
import importlib
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import bench

P = importlib.import_module("scilmm_amd.SparseCholesky")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "100k"
    t0 = time.time()
    A, C, y = bench.build_problem(name, 0)
    print("problem: n=%d nnz=%d  (%.1f s)" % (A.shape[0], A.nnz, time.time() - t0))
    chol = P.SparseCholesky()
    log = []
    orig = P.bolt_gradient_estimation

    def rec(x, *a, **k):
        t = time.time()
        out = orig(x, *a, **k)
        log.append(time.time() - t)
        return out

    P.bolt_gradient_estimation = rec
    np.random.seed(1)
    t0 = time.time()
    res = P.REML(chol, [A], C, y)
    tot = time.time() - t0
    print("REML: %.1f s total, %d evaluations, first %.2f s (includes symbolic), median later %.3f s" %
          (tot, len(log), log[0], float(np.median(log[1:])) if len(log) > 1 else 0))
    print("sigma2", res["covariance coefficients"], "beta", res["covariates coefficients"], "std", res["covariance std"])
    print("time outside evaluations (HE start, final factor, std-errs): %.1f s" % (tot - sum(log)))
    # split of one evaluation
    mats = [A, sp.eye(A.shape[0]).tocsr()]
    sym = chol.engine_for(mats)
    ys = y / y.std()
    n = A.shape[0]
    t = time.time(); R = np.random.randn(n, 100); t_rng = time.time() - t
    fac = sym.factorize([0.4, 0.6])
    t = time.time(); fac.refactorize([0.41, 0.59]); t_fac = time.time() - t
    t = time.time(); Z = fac.lmul(R); t_lmul = time.time() - t
    t = time.time(); X = fac(np.hstack([C, ys[:, None], Z])); t_solve = time.time() - t
    t = time.time(); q = sym.quadforms(0, X); t_q0 = time.time() - t
    t = time.time(); q = sym.quadforms(1, X); t_q1 = time.time() - t
    print("one evaluation, host-visible: rng %.3f  refactorize %.3f  lmul %.3f  solve(103) %.3f  quad(A) %.3f  quad(I) %.3f s" %
          (t_rng, t_fac, t_lmul, t_solve, t_q0, t_q1))
    print("device timers:", {k: round(v, 2) for k, v in sym.timing().items()})


if __name__ == "__main__":
    main()
