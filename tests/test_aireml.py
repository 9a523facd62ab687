"""AI-REML: the branch the reference leaves as a stub (scilmm/SparseCholesky.py:128-130), built on the
same evaluation as the L-BFGS-B path.  CPU part: the iteration itself on a dense stand-in factor; GPU part: the same fit
through the HIP engine."""
import importlib

import numpy as np
import pytest
import scipy.optimize as opt
import scipy.sparse as sp


def _problem():
    from scilmm_amd.harness.pedigree import make_problem
    mats, C, y = make_problem(1500, 0.01, seed=3)
    return mats[0], C, y


def test_ai_reml_reaches_the_optimum_of_the_same_objective():
    """With common random numbers the Monte-Carlo estimator is a smooth function of sigma2: AI-REML must land next to the
    point a tightly converged L-BFGS-B finds for that same function, in a handful of Newton-type steps, drive the
    estimated gradient to zero when iterated further, and advance the global RNG stream by exactly one evaluation."""
    from oracle import oracle as O
    P = importlib.import_module("scilmm_amd.SparseCholesky")
    A, C, y = _problem()
    n = A.shape[0]
    chol = lambda V: O.DenseFactor(V)
    evals = []
    orig = P.bolt_gradient_estimation

    def counting(*a, **k):
        evals.append(1)
        return orig(*a, **k)

    P.bolt_gradient_estimation = counting
    try:
        np.random.seed(1)
        got = P.REML(chol, [A], C, y, sim_num=200, aireml=True)
    finally:
        P.bolt_gradient_estimation = orig
    after = np.random.randn()
    np.random.seed(1)
    np.random.randn(n, 200)
    assert after == np.random.randn()                       # the stream moved by one evaluation's draw
    assert len(evals) <= 12
    ys = y / y.std()
    mm = [A, sp.eye(n).tocsr()]
    np.random.seed(1)
    st = np.random.get_state()

    def f(x):
        np.random.set_state(st)
        return orig(x, chol, mm, C, ys, True, 200, False, True)

    res = opt.minimize(f, np.log([0.3, 0.7]), jac=True, method="L-BFGS-B", options={"ftol": 1e-13, "gtol": 1e-10})
    best = np.exp(res.x)
    # at the reference's tolerance (relative likelihood change 1e-7) the flat likelihood leaves ~1 % in sigma2 ...
    assert np.abs(got["covariance coefficients"] - best).max() < 2e-2 * best.max()
    assert np.all(got["covariance std"] > 0)
    # ... and iterated to a tight tolerance AI-REML sits on the optimum itself
    np.random.set_state(st)
    tight = P._ai_reml(chol, mm, C, ys, np.array([0.3, 0.7]), True, 200, False, ftol=1e-14, tol=1e-10)
    # (the estimator's gradient is not the derivative of its likelihood value -- exact log-det, Monte-Carlo trace -- so a
    #  line-search method that mixes the two stops near, not at, the gradient's root; AI-REML iterates on the root itself)
    g_tight = np.abs(f(np.log(tight))[1]).max()
    g_lbfgs = np.abs(f(np.log(best))[1]).max()
    assert g_tight < 1e-8 * abs(res.fun) and g_tight <= g_lbfgs
    assert np.abs(tight - best).max() < 1e-2 * best.max()


@pytest.mark.gpu
def test_ai_reml_on_the_hip_engine_matches_the_dense_stand_in():
    """Same fit, same permutation (identity), same np.random stream: the HIP engine and the dense LAPACK stand-in walk the
    same AI-REML iterates -- sigma2 to 1e-6 (north_star's bar for the default optimiser, met by this one as well)."""
    from oracle import oracle as O
    P = importlib.import_module("scilmm_amd.SparseCholesky")
    A, C, y = _problem()
    n = A.shape[0]
    np.random.seed(2)
    ref = P.REML(lambda V: O.DenseFactor(V), [A], C, y, aireml=True)
    np.random.seed(2)
    got = P.REML(P.SparseCholesky(perm=np.arange(n)), [A], C, y, aireml=True)
    for key, tol in (("covariance coefficients", 1e-6), ("covariates coefficients", 1e-5), ("covariance std", 1e-5)):
        assert np.abs(got[key] - ref[key]).max() < tol * np.abs(ref[key]).max(), key
