"""Ordering / fill study (CPU only): the engine's approximate minimum degree vs its nested
dissection (default separator-acceptance threshold, and always-dissect) vs SciPy SuperLU's MMD_AT_PLUS_A, on the
simulated pedigrees of the BASELINE configs.  Reports nnz(L), sum colcount^2 (factor flops, CHOLMOD's `fl`), the
largest column count (= width of the top clique) and, as a lower bound no ordering can beat, the largest clique the
relatedness graph is known to contain: the descendants of one individual are pairwise related.

    python tools/ordering_study.py profiles/r2_ordering.json 10k:10000:0.001:1 30k:30000:0.005:1 100k:100000:0.005:0
(name:n:sparsity_factor:run_superlu; SuperLU factorizes to produce its ordering -- affordable up to ~30k).
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scilmm_amd import _lib  # noqa: E402
from scilmm_amd.harness.pedigree import simulate_pedigree, ibd_from_parents, drop_unrelated  # noqa: E402


def max_descendant_clique(par, has):
    """Largest descendant set (incl. the ancestor) among the kept individuals: a clique of the relatedness graph."""
    n = par.shape[0]
    depth = np.zeros(n, dtype=np.int64)
    while True:
        da = np.where(par[:, 0] >= 0, depth[np.maximum(par[:, 0], 0)] + 1, 0)
        db = np.where(par[:, 1] >= 0, depth[np.maximum(par[:, 1], 0)] + 1, 0)
        nd = np.maximum(da, db)
        if np.array_equal(nd, depth):
            break
        depth = nd
    S = sp.identity(n, format="csr", dtype=np.float32)
    for d in range(1, int(depth.max()) + 1):
        rows = np.where(depth == d)[0]
        r = np.repeat(rows, 2)
        p = par[rows].reshape(-1)
        k = p >= 0
        H = sp.csr_matrix((np.ones(k.sum(), np.float32), (r[k], p[k])), shape=(n, n))
        S = S + H @ S
        S.data[:] = 1
    S = S[has]  # descendants that survive the unrelated-drop
    return int(np.asarray(S.sum(axis=0)).max())


def main():
    out_path = sys.argv[1]
    cases = [tuple(x.split(":")) for x in sys.argv[2:]]
    res = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for name, n0, sf, mmd in cases:
        n0, sf = int(n0), float(sf)
        t = time.time()
        par, sex, gen = simulate_pedigree(n0, sf, 0)
        A, has = drop_unrelated(ibd_from_parents(par))[:2]
        n = A.shape[0]
        row = {"n_individuals": n0, "sparsity_factor": sf, "n_after_unrelated_drop": n, "nnz_A": int(A.nnz),
               "generate_s": time.time() - t, "orderings": {}}
        row["lower_bound_max_clique"] = max_descendant_clique(par, has)
        print(name, "largest descendant clique", row["lower_bound_max_clique"], flush=True)
        for m in ("amd", "nesdis", "nesdis_always"):
            t = time.time()
            p = _lib.order(A, m)
            dt = time.time() - t
            nz, fl, mx = _lib.fill_count(A, p)
            row["orderings"][m] = {"nnzL": nz, "sum_cc2": fl, "max_colcount": mx, "order_s": dt}
            print(name, m, nz, fl, mx, "%.1fs" % dt, flush=True)
        if int(mmd):
            V = (0.4 * A + 0.6 * sp.identity(n)).tocsc()
            t = time.time()
            lu = sla.splu(V, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options={"SymmetricMode": True})
            dt = time.time() - t
            perm = np.argsort(lu.perm_c).astype(np.int32)
            nz, fl, mx = _lib.fill_count(A, perm)
            row["orderings"]["superlu_mmd_at_plus_a"] = {"nnzL": nz, "sum_cc2": fl, "max_colcount": mx,
                                                         "order_and_factor_s": dt}
            print(name, "mmd", nz, fl, mx, "%.1fs" % dt, flush=True)
        res[name] = row
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
