"""CPU tests: the oracle (oracle/) against LAPACK, SuperLU and the golden vectors made by the
reference's own Python (tests/golden, oracle/make_golden.py).  No GPU needed."""
import os

import numpy as np
import scipy.sparse as sp

from oracle import oracle as O
from oracle import reml_oracle as RO
from tests.helpers import random_spd, rel_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _g1():
    g = np.load(os.path.join(GOLD, "G1_reml_2000.npz"))
    A = sp.csr_matrix((g["A_data"], g["A_indices"], g["A_indptr"]), shape=tuple(g["A_shape"]))
    return g, A


def test_c_oracle_vs_lapack_and_superlu():
    rng = np.random.default_rng(1)
    for trial in range(8):
        n = int(rng.integers(3, 160))
        V = random_spd(n, float(rng.uniform(0.02, 0.3)), trial)
        for P in (None, rng.permutation(n)):
            f, d, s = O.OracleFactor(V, P), O.DenseFactor(V, P), O.SuperLUFactor(V)
            b = rng.standard_normal((n, 4))
            assert rel_err(f(b), d(b)) < 1e-11
            assert rel_err(f(b), s(b)) < 1e-9
            assert abs(f.logdet() - d.logdet()) < 1e-10 * max(1, abs(d.logdet()))
            assert abs(f.logdet() - s.logdet()) < 1e-9 * max(1, abs(d.logdet()))
            assert rel_err(f.L().toarray(), d.L().toarray()) < 1e-12
            R = rng.standard_normal((n, 3))
            assert rel_err(f.lmul(R), (d.L() @ R)[np.argsort(d.P())]) < 1e-12
        U = rng.standard_normal((n, 5))
        assert rel_err(O.quadforms(V, U), ((V @ U) * U).sum(axis=0)) < 1e-12
        assert rel_err(O.quadforms(sp.tril(V).tocsr(), U, lower_only=True), ((V @ U) * U).sum(axis=0)) < 1e-12


def test_oracle_not_positive_definite():
    import pytest
    with pytest.raises(O.NotPositiveDefinite):
        O.OracleFactor(sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 1.0]])))


def test_golden_G0_reference_fixture():
    """V = A/2 + I/2 on the reference's only deterministic fixture (relationship_example.csv)."""
    g = np.load(os.path.join(GOLD, "G0_relationship_example.npz"))
    A = g["A"]
    # known dyadic values quoted in SURVEY.md section 4
    assert A[9, 9] == 1.0625 and A[6, 7] == 0.125 and A[6, 9] == 0.5625 and A[7, 9] == 0.5625
    V = sp.csr_matrix(g["V"])
    f = O.OracleFactor(V)
    assert rel_err(f.L().toarray(), g["chol"]) < 1e-13
    assert abs(f.logdet() - float(g["logdet"])) < 1e-12
    assert rel_err(f(np.eye(10)), g["Vinv"]) < 1e-12


def test_supernodal_cpu_baseline_matches_oracle():
    from scilmm_amd.factor import Symbolic
    A = random_spd(400, 0.03, 5)
    n = A.shape[0]
    sym = Symbolic([A, sp.identity(n, format="csr")], upload=False)
    cpu = O.SupernodalCPU(sym.arrays(), n)
    perm = sym.get("perm")
    Lw = sp.tril(A[perm][:, perm]).tocsc()
    Lw.sort_indices()
    vals = 0.7 * Lw.data
    vals[Lw.indptr[:-1]] += 0.3
    cpu.assemble(vals)
    cpu.factorize()
    o = O.OracleFactor((0.7 * A + 0.3 * sp.identity(n)).tocsr(), perm)
    assert abs(cpu.logdet() - o.logdet()) < 1e-10 * abs(o.logdet())
    B = np.random.default_rng(0).standard_normal((n, 7))
    assert rel_err(cpu.solve(B), o(B)) < 1e-10


def test_reml_oracle_single_evaluation_vs_reference_golden():
    g, A = _g1()
    n = A.shape[0]
    y = g["y"] / g["y"].std()
    mats = [A, sp.eye(n).tocsr()]
    np.random.seed(3)
    nll, grad = RO.evaluate(np.log([0.3, 0.7]), mats, g["C"], y, reml=False, sim_num=100, perm=g["amd_perm"])
    assert abs(nll - float(g["ml_nll"])) < 1e-10 * abs(float(g["ml_nll"]))
    assert rel_err(grad, g["ml_grad"]) < 1e-8
    assert rel_err(RO.he([A], g["C"], y), g["he"]) < 1e-10


def _check_traj(trace, g, tag, n_check=None):
    m = min(len(trace), len(g["%s_nll" % tag])) if n_check is None else n_check
    for i in range(m):
        x, nll, grad = trace[i]
        assert rel_err(x, g["%s_x" % tag][i]) < 1e-6, (tag, i)
        assert abs(nll - g["%s_nll" % tag][i]) < 1e-7 * abs(g["%s_nll" % tag][i]), (tag, i)
        assert rel_err(grad, g["%s_grad" % tag][i]) < 1e-5, (tag, i)


def test_reml_oracle_fit_reproduces_reference_trajectory():
    """Same P + same RNG stream => same trajectory as the reference's REML (sigma2 within 1e-6)."""
    g, A = _g1()
    for tag in ("amd", "ident"):
        trace = []
        np.random.seed(1)
        s2, beta, std = RO.fit([A], g["C"], g["y"].copy(), perm=g["%s_perm" % tag], trace=trace)
        # the stopping test sits on the Monte-Carlo noise floor: allow the evaluation COUNT to differ by a
        # couple of trailing evaluations, compare the common prefix and the final estimates
        assert abs(len(trace) - len(g["%s_nll" % tag])) <= 2
        _check_traj(trace, g, tag)
        assert rel_err(s2, g["%s_sigma2" % tag]) < 1e-6
        assert rel_err(beta, g["%s_beta" % tag]) < 1e-6
        assert rel_err(std, g["%s_std" % tag]) < 1e-6


def test_product_host_logic_with_oracle_factor_reproduces_reference():
    """The product's REML driver is factor-agnostic (like the reference): fed the oracle factor on the CPU it
    must reproduce the reference trajectory too.  This checks the HOST logic; the HIP path is in -m gpu."""
    import importlib
    P = importlib.import_module("scilmm_amd.SparseCholesky")  # the attribute of the package is the class
    g, A = _g1()
    perm = g["amd_perm"]
    trace = []
    orig = P.bolt_gradient_estimation

    def rec(x, *a, **k):
        nll, grad = orig(x, *a, **k)
        trace.append((np.array(x), nll, np.array(grad)))
        return nll, grad

    P.bolt_gradient_estimation = rec
    try:
        np.random.seed(1)
        res = P.REML(lambda V: O.OracleFactor(V, perm), [A], g["C"], g["y"].copy())
    finally:
        P.bolt_gradient_estimation = orig
    _check_traj(trace, g, "amd")
    assert rel_err(res["covariance coefficients"], g["amd_sigma2"]) < 1e-6
    assert rel_err(res["covariates coefficients"], g["amd_beta"]) < 1e-6
    assert rel_err(res["covariance std"], g["amd_std"]) < 1e-6
    np.random.seed(5)
    he_est, he_std = P.HE([A], g["C"], g["y"] / g["y"].std(), compute_stderr=True)
    assert rel_err(he_est, g["he_est"]) < 1e-10 and rel_err(he_std, g["he_std"]) < 1e-8


def test_product_lmm_with_oracle_factor_reproduces_reference_K3():
    import importlib
    M = importlib.import_module("scilmm_amd.Estimation.LMM")
    g, A = _g1()
    g2 = np.load(os.path.join(GOLD, "G2_lmm_dominance.npz"))
    D = sp.csr_matrix((g2["D_data"], g2["D_indices"], g2["D_indptr"]), shape=A.shape)
    perm = g2["perm"]
    np.random.seed(2)
    res = M.LMM(lambda V: O.OracleFactor(V, perm), [A, D], g2["cov"], g["y"].copy())
    assert rel_err(res["covariance coefficients"], g2["sigma2"]) < 1e-5
    assert rel_err(res["covariates coefficients"], g2["beta"]) < 1e-5
    assert rel_err(res["covariance std"], g2["std"]) < 1e-4
    assert rel_err(res["covariates p-values"], g2["pvalues"]) < 1e-5


def test_dominance_restatement_matches_the_reference_golden():
    """oracle.dominance against the dominance matrix the reference's own code produced (tests/golden/G2, made by
    oracle/make_golden.py from scilmm.Matrices.Dominance.dominance): same pattern, same values to the last bit."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g1 = np.load(os.path.join(gold, "G1_reml_2000.npz"))
    g2 = np.load(os.path.join(gold, "G2_lmm_dominance.npz"))
    shape = tuple(g1["A_shape"])
    A = sp.csr_matrix((g1["A_data"], g1["A_indices"], g1["A_indptr"]), shape=shape)
    rel = sp.csr_matrix((g2["rel_data"], g2["rel_indices"], g2["rel_indptr"]), shape=shape)
    Dref = sp.csr_matrix((g2["D_data"], g2["D_indices"], g2["D_indptr"]), shape=shape)
    Dref.sort_indices()
    from scilmm_amd.Matrices.Dominance import parents_of
    Dm = O.dominance(parents_of(rel), A)
    assert np.array_equal(Dm.indptr, Dref.indptr) and np.array_equal(Dm.indices, Dref.indices)
    assert np.array_equal(Dm.data, Dref.data)
