"""ORACLE -- TEST INFRASTRUCTURE ONLY.

NumPy restatement of the reference's REML evaluation and fit, written independently of the product's
host code (scilmm_amd/SparseCholesky.py) so that the two can be compared:

* ``evaluate``  follows reference scilmm/SparseCholesky.py:77-117 (bolt_gradient_estimation) with its helpers
  :29-34 (GLS fixed effects), :37-46 (negative log-likelihood), :49-52 (simulated vectors, global legacy RNG),
  :55-59 (weighted sum), :62-74 (Monte-Carlo gradient);
* ``fit``       follows :120-144 (HE start + L-BFGS-B with eps=1e-5, ftol=1e-7), :147-174 (Hessian / std-errs) and
  :177-189 (REML wrapper: y / y.std(), identity appended);
* ``he``        follows :192-246 (HE point estimate).

Pinned against tests/golden/G1_reml_2000.npz (the reference's own Python driven by a dense factor) in
tests/test_oracle.py.  The factor is any object with the four-member protocol; by default the C oracle.
"""
import numpy as np
import scipy.linalg as la
import scipy.optimize as opt
import scipy.sparse as sp

from . import oracle as O


def weighted_sum(mats, s2):
    V = None
    for a, m in zip(s2, mats):
        V = a * m if V is None else V + a * m
    return V.tocsc()


def evaluate(log_s2, mats, C, y, reml=True, sim_num=100, perm=None, factor_of=None, take_exp=True):
    """(nll, grad wrt log sigma2) at one point; consumes n*sim_num normals from np.random."""
    s2 = np.exp(log_s2) if take_exp else np.asarray(log_s2, dtype=float)
    V = weighted_sum(mats, s2)
    n = V.shape[0]
    f = factor_of(V) if factor_of is not None else O.OracleFactor(V, perm)
    pinv = np.argsort(f.P())
    ViC = f(C)
    G = la.cho_factor(C.T @ ViC)                     # C' V^-1 C
    beta = la.cho_solve(G, C.T @ f(y))
    resid = y - C @ beta
    Viy = f(resid)
    nll = 0.5 * (resid @ Viy + n * np.log(2.0 * np.pi) + f.logdet())
    if reml:
        nll += np.log(np.diag(G[0])).sum()
    R = np.random.randn(n, sim_num)
    # (a factor that offers the fused product is asked for it: exporting L at the 100k config is 2 GB of CSC)
    Z = f.lmul(R) if hasattr(f, "lmul") else (f.L() @ R)[pinv]
    U = f(Z)
    g = np.empty(len(s2))
    for k, Ak in enumerate(mats):
        tr_hat = ((Ak @ U) * U).sum(axis=0).mean()   # ~ tr(V^-1 A_k)
        g[k] = 0.5 * (tr_hat - Viy @ (Ak @ Viy))
        if reml:
            g[k] -= 0.5 * np.trace(la.cho_solve(G, ViC.T @ (Ak @ ViC)))
    return nll, (g * s2 if take_exp else g)


def he(mats, C, y):
    yr = y - C @ np.linalg.solve(C.T @ C, C.T @ y)
    yr = yr / yr.std()
    K = len(mats)
    q = np.array([yr @ (m @ yr) - m.diagonal() @ yr ** 2 for m in mats])
    S = np.empty((K, K))
    for i in range(K):
        for j in range(i + 1):
            S[i, j] = S[j, i] = mats[i].multiply(mats[j]).sum() - mats[i].diagonal() @ mats[j].diagonal()
    return np.linalg.solve(S, q)


def fit(mats, C, y, reml=True, sim_num=100, perm=None, factor_of=None, trace=None):
    """REML fit; returns (sigma2, beta, std).  ``trace`` (a list) receives (x, nll, grad) per evaluation."""
    y = y / y.std()
    n = y.size
    mats = list(mats) + [sp.eye(n).tocsr()]
    h = he(mats[:-1], C, y)
    x0 = np.concatenate([h, [1.0 - h.sum()]])
    if np.any(x0 < 0):
        x0 = np.ones(len(mats))
    x0 = x0 / x0.sum()

    def fun(x):
        nll, g = evaluate(x, mats, C, y, reml, sim_num, perm, factor_of)
        if trace is not None:
            trace.append((np.array(x), nll, np.array(g)))
        return nll, g

    res = opt.minimize(fun, np.log(x0), jac=True, method="L-BFGS-B", options={"eps": 1e-5, "ftol": 1e-7})
    s2 = np.exp(res.x)
    V = weighted_sum(mats, s2)
    f = factor_of(V) if factor_of is not None else O.OracleFactor(V, perm)
    ViC = f(C)
    G = la.cho_factor(C.T @ ViC)
    beta = la.cho_solve(G, C.T @ f(y))

    def proj(z):
        Viz = f(z)
        return Viz - ViC @ la.cho_solve(G, C.T @ Viz)

    K = len(mats)
    Py = proj(y)
    H = np.empty((K, K))
    for j in range(K):
        t = proj(mats[j] @ Py)
        for i in range(j + 1):
            H[i, j] = H[j, i] = -0.5 * (y @ proj(mats[i] @ t))
    std = np.sqrt(np.diag(la.inv(-H)) * (1.0 + 1.0 / sim_num))
    return s2, beta, std
