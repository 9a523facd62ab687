"""Generates tests/golden/*.npz by running the REFERENCE's own Python (imported read-only from
/root/reference) with a dense LAPACK stand-in for the absent scikit-sparse factor.

Run in the build container only:   python oracle/make_golden.py
(the reference never travels to the GPU box; only the arrays written here do.)

Why this pins parity although CHOLMOD is absent: for a fixed permutation P the Cholesky factor of
V[P][:,P] is unique, so any correct factor -- CHOLMOD, LAPACK, the C oracle or the HIP engine -- fed with
the same P and the same ``np.random`` stream produces the same nll / gradient / sigma2 trajectory up
to fp64 rounding.  The stand-in implements exactly the four members the reference touches
(reference scilmm/SparseCholesky.py:30,32,40,50,52,93,100).

Local accommodations applied from here (the reference is never edited): ``np.float``/``np.bool``
aliases removed in NumPy >= 1.24 (SparseCholesky.py:384, Simulation/Pedigree.py:100) and an
object-array fallback for the ragged ``np.array`` in Simulation/Pedigree.py:44,54.
"""
import os
import sys
import tempfile
import types

import numpy as np
import scipy.linalg as la
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


class StandInFactor(object):
    current_perm = None  # set by the driver before each fit

    def __init__(self, V):
        Vd = V.toarray()
        n = Vd.shape[0]
        self._P = np.arange(n) if StandInFactor.current_perm is None else np.asarray(StandInFactor.current_perm)
        self._L = la.cholesky(Vd[np.ix_(self._P, self._P)], lower=True)

    def __call__(self, b):
        x = la.cho_solve((self._L, True), np.asarray(b)[self._P])
        out = np.empty_like(x)
        out[self._P] = x
        return out

    def L(self):
        return sp.csc_matrix(np.tril(self._L))

    def P(self):
        return self._P

    def logdet(self):
        return 2.0 * np.log(np.diag(self._L)).sum()


def import_reference():
    m = types.ModuleType("sksparse")
    c = types.ModuleType("sksparse.cholmod")
    c.cholesky = lambda A, **kw: StandInFactor(A)
    m.cholmod = c
    sys.modules["sksparse"] = m
    sys.modules["sksparse.cholmod"] = c
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "bool"):
        np.bool = bool
    sys.path.insert(0, REF)
    import scilmm  # noqa: F401
    import scilmm.Estimation.LMM  # noqa: F401
    ped = sys.modules["scilmm.Simulation.Pedigree"]

    class NpProxy(object):
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def array(obj, *a, **k):
            try:
                return np.array(obj, *a, **k)
            except ValueError:
                out = np.empty(len(obj), dtype=object)
                for i, o in enumerate(obj):
                    out[i] = o
                return out

    ped.np = NpProxy()
    return sys.modules["scilmm.SparseCholesky"], sys.modules["scilmm.Estimation.LMM"]


def record_trajectory(mod, name="bolt_gradient_estimation"):
    orig = getattr(mod, name)
    log = []

    def wrapped(x, *a, **k):
        nll, grad = orig(x, *a, **k)
        log.append((np.array(x, dtype=float), float(nll), np.array(grad, dtype=float)))
        return nll, grad

    setattr(mod, name, wrapped)
    return log, orig


def engine_perm(mats):
    from scilmm_amd.factor import Symbolic
    n = mats[0].shape[0]
    S = Symbolic(list(mats) + [sp.identity(n, format="csr")], upload=False)
    return S.P()


def main():
    os.makedirs(OUT, exist_ok=True)
    refmod, lmmmod = import_reference()
    from scilmm.Matrices.Numerator import simple_numerator
    from scilmm.Matrices.Dominance import dominance
    from scilmm.Simulation.Pedigree import simulate_tree
    from scilmm.FileFormats.pedigree import Pedigree
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        # ---- G0: the reference's only deterministic fixture (scilmm/Tests/Examples/relationship_example.csv)
        fam_path = os.path.join(REF, "scilmm", "Tests", "Examples", "relationship_example.csv")
        ped = Pedigree({})
        ped.load_pedigree(fam_path)
        ped.compute_all_values()
        rel = ped.relationship
        A0, L0, D0 = simple_numerator(rel)
        ids = np.array(list(ped.entries_dict.values()))
        V0 = (0.5 * A0 + 0.5 * sp.identity(10)).toarray()
        Lc = la.cholesky(V0, lower=True)
        np.savez(os.path.join(OUT, "G0_relationship_example.npz"),
                 fam_text=np.array(open(fam_path).read()), ids=ids,
                 rel=rel.toarray().astype(np.int8), A=A0.toarray(), L=L0.toarray(), D=D0.toarray(),
                 V=V0, chol=Lc, logdet=2 * np.log(np.diag(Lc)).sum(), Vinv=np.linalg.inv(V0))

        # ---- G1: seeded simulated pedigree, full REML trajectory (K=2), identity P and engine AMD P
        np.random.seed(7)
        rel, sex, gen = simulate_tree(2000, 0.01, 1.4, 0.8)
        A, L, D = simple_numerator(rel)
        A = sp.csr_matrix(A)
        has = np.asarray(A.sum(axis=1)).ravel() > 1
        A = A[has][:, has].tocsr()
        A.eliminate_zeros()
        A.sort_indices()
        n = A.shape[0]
        relh = sp.csr_matrix(rel)[has][:, has]
        sexh = sex[has]
        np.random.seed(11)
        Lp = sp.csr_matrix(L)[has][:, has]
        g = Lp.dot(np.random.randn(n))
        e = np.random.randn(n)
        y = np.sqrt(0.4) * (g - g.mean()) / g.std() + np.sqrt(0.6) * (e - e.mean()) / e.std() + 0.01 * sexh
        y = (y - y.mean()) / y.std()
        C = np.stack([(sexh - sexh.mean()) / sexh.std(), np.ones(n)], axis=1)
        out = dict(A_data=A.data, A_indices=A.indices, A_indptr=A.indptr, A_shape=np.array(A.shape), y=y, C=C)
        perm_amd = engine_perm([A])
        for tag, perm in (("ident", None), ("amd", perm_amd)):
            StandInFactor.current_perm = perm
            log, orig = record_trajectory(refmod)
            np.random.seed(1)
            res = refmod.REML(refmod.SparseCholesky(), [A], C, y.copy(), verbose=False)
            refmod.bolt_gradient_estimation = orig
            out["%s_perm" % tag] = np.arange(n) if perm is None else perm
            out["%s_x" % tag] = np.array([l[0] for l in log])
            out["%s_nll" % tag] = np.array([l[1] for l in log])
            out["%s_grad" % tag] = np.array([l[2] for l in log])
            out["%s_sigma2" % tag] = res["covariance coefficients"]
            out["%s_beta" % tag] = res["covariates coefficients"]
            out["%s_std" % tag] = res["covariance std"]
            print("G1", tag, "evals", len(log), "sigma2", res["covariance coefficients"], "beta",
                  res["covariates coefficients"], "std", res["covariance std"])
        # ML (reml=False) single evaluation and HE start
        StandInFactor.current_perm = perm_amd
        np.random.seed(3)
        mats = [A, sp.eye(n).tocsr()]
        ys = y / y.std()
        nll_ml, grad_ml = refmod.bolt_gradient_estimation(np.log([0.3, 0.7]), refmod.SparseCholesky(), mats, C, ys,
                                                          False, 100, False)
        out["ml_nll"] = nll_ml
        out["ml_grad"] = grad_ml
        out["he"] = refmod.HE([A], C, ys.copy(), compute_stderr=False)
        np.random.seed(5)
        he2 = refmod.HE([A], C, ys.copy(), compute_stderr=True)
        out["he_est"], out["he_std"] = he2
        np.savez_compressed(os.path.join(OUT, "G1_reml_2000.npz"), **out)

        # ---- G2: three components (A, dominance, I) through the legacy LMM entry point
        Dm = sp.csr_matrix(dominance(relh, A.copy()))
        Dm.sort_indices()
        perm3 = engine_perm([A, Dm])
        StandInFactor.current_perm = perm3
        log, orig = record_trajectory(lmmmod)
        np.random.seed(2)
        cov1 = C[:, :1].copy()
        res = lmmmod.LMM(lmmmod.SparseCholesky(), [A, Dm], cov1, y.copy(), verbose=False)
        lmmmod.bolt_gradient_estimation = orig
        print("G2 evals", len(log), res)
        np.savez_compressed(os.path.join(OUT, "G2_lmm_dominance.npz"),
                            D_data=Dm.data, D_indices=Dm.indices, D_indptr=Dm.indptr, perm=perm3, cov=cov1,
                            x=np.array([l[0] for l in log]), nll=np.array([l[1] for l in log]),
                            grad=np.array([l[2] for l in log]), sigma2=res["covariance coefficients"],
                            beta=res["covariates coefficients"], std=res["covariance std"],
                            pvalues=res["covariates p-values"],
                            rel_data=relh.data.astype(np.int8), rel_indices=relh.indices, rel_indptr=relh.indptr)
    finally:
        os.chdir(cwd)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
