"""Second entry point onto the same factor boundary: reference ``scilmm/Estimation/LMM.py``.

``LMM(cholesky, mats, covariates, y, with_intercept=True, reml=True, sim_num=100, verbose=False)``
(LMM.py:154-169) differs from ``REML`` only in its conventions: the intercept is PREPENDED (:156-157),
the optimiser starts from equal variance shares (:113-116), y is used as given (no rescaling), fixed-effect
p-values come from F(1, n-1) (:129-133) and the standard errors use the un-projected recursion of
``compute_sig_of_sig`` (:136-151).  The per-evaluation work is the shared device path of
``scilmm_amd.SparseCholesky.bolt_gradient_estimation``.
"""
from itertools import combinations_with_replacement

import numpy as np
import scipy.linalg as la
import scipy.optimize as optimize
import scipy.stats as stats
from scipy.sparse import eye

from ..SparseCholesky import (SparseCholesky, _final_factor, bolt_gradient_estimation as _bolt,  # noqa: F401
                              estimate_fixed_effects as compute_fixed_effects, negative_log_likelihood,
                              simulate_vector, matrices_weighted_sum, compute_gradients)

np.set_printoptions(precision=4, linewidth=200)


def bolt_gradient_estimation(log_sig2g_array, cholesky, mats, covariates, y, reml, sim_num, verbose):
    """LMM.py:75-110 (always in log space)."""
    return _bolt(log_sig2g_array, cholesky, mats, covariates, y, reml, sim_num, verbose, True)


def compute_sigmas(cholesky, mats, covariates, y, reml=True, sim_num=100, verbose=True):
    """L-BFGS-B from equal shares (LMM.py:113-126)."""
    x0 = np.full(len(mats), 1.0 / len(mats))
    res = optimize.minimize(bolt_gradient_estimation, np.log(x0),
                            args=(cholesky, mats, covariates, y, reml, sim_num, verbose),
                            jac=True, method='L-BFGS-B', options={'eps': 1e-5, 'ftol': 1e-7})
    return np.exp(res.x)


def compute_fixed_effects_p_value(y, covariates, fixed_effects, L_CT_invV_C):
    """Wald-type F(1, n-1) p-values (LMM.py:129-133)."""
    var_fe = la.cho_solve(L_CT_invV_C, np.eye(covariates.shape[1]))
    return stats.f(1, y.shape[0] - 1).sf(fixed_effects ** 2 / np.diag(var_fe))


def compute_sig_of_sig(mats, covariates, factor, y, sim_num):
    """Standard errors of the variance components (LMM.py:136-151)."""
    K = len(mats)
    from ..SparseCholesky import _DeviceProjector, _hip_factor_of, _spmm_on_device
    hip = _hip_factor_of(factor, mats)
    if hip is not None:
        # the same recursion with every n-vector in HBM: V^-1 [C | y] (one sweep), K SpMMs on P y, one K-column sweep, K SpMMs
        # on K columns, one K^2-column sweep (the reference's form: K (K + 3) / 2 + 2 single-column host round trips)
        torch, sym = hip
        pr = _DeviceProjector(torch, sym, factor, covariates, y)
        Py_d = pr.project_solved(pr.Viy[:, None])
        inner_d = pr.solve(torch.cat([_spmm_on_device(torch, sym, j, Py_d) for j in range(K)], dim=1))      # column j = V^-1 A_j P y
        cols = pr.solve(torch.cat([_spmm_on_device(torch, sym, i, inner_d) for i in range(K)], dim=1))      # column i K + j
        v = (torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).cuda() @ cols).cpu().numpy().reshape(K, K)
        hess = np.empty((K, K))
        for i, j in combinations_with_replacement(range(K), 2):
            hess[i, j] = hess[j, i] = -0.5 * v[i, j]
        return np.sqrt(np.diag(la.inv(-hess) * (1 + 1.0 / sim_num)))
    V_inv_y = factor(y)
    V_inv_C = factor(covariates)
    CtViC_inv = np.linalg.inv(covariates.T.dot(V_inv_C))
    Py = V_inv_y - V_inv_C.dot(CtViC_inv.dot(covariates.T.dot(V_inv_y)))
    inner = [factor(mats[j].dot(Py)) for j in range(K)]
    hess = np.empty((K, K))
    for i, j in combinations_with_replacement(range(K), 2):
        hess[i, j] = hess[j, i] = -0.5 * y.T.dot(factor(mats[i].dot(inner[j])))
    return np.sqrt(np.diag(la.inv(-hess) * (1 + 1.0 / sim_num)))


def LMM(cholesky, mats, covariates, y, with_intercept=True, reml=True, sim_num=100, verbose=False):
    mats = list(mats) + [eye(y.size).tocsr()]
    if with_intercept:
        covariates = np.hstack((np.ones((y.size, 1)), covariates))
    mats_coefficients = compute_sigmas(cholesky, mats, covariates, y, reml, sim_num, verbose)
    factor = _final_factor(cholesky, mats, mats_coefficients)
    _, L_CT_invV_C, _, fixed_effects = compute_fixed_effects(factor, y, covariates)
    p_values = compute_fixed_effects_p_value(y, covariates, fixed_effects, L_CT_invV_C)
    sigmas_sigmas = compute_sig_of_sig(mats, covariates, factor, y, sim_num)
    del factor
    if isinstance(cholesky, SparseCholesky):
        cholesky.release_factors()
    return {"covariance coefficients": mats_coefficients,
            "covariates coefficients": fixed_effects,
            "covariance std": sigmas_sigmas,
            "covariates p-values": p_values}
