"""Shared test helpers: seeded SPD problems and oracle comparisons."""
import numpy as np
import scipy.sparse as sp


def random_spd(n, density, seed):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=int(rng.integers(1 << 30)), format="csr")
    M = (M + M.T).tocsr()
    d = np.asarray(abs(M).sum(axis=1)).ravel() + 1.0 + rng.uniform(0, 1, n)
    A = (M + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


def small_pedigree(n, sf, seed=0):
    from scilmm_amd.harness.pedigree import simulate_pedigree, ibd_from_parents, drop_unrelated
    par, sex, _ = simulate_pedigree(n, sf, seed)
    A = ibd_from_parents(par)
    A2, has, sex2 = drop_unrelated(A, sex)
    return A2, sex2


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
