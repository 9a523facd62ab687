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


def small_pedigree_k3(n, sf, seed=0):
    """(A, D, sex) after the unrelated-drop: additive + dominance matrix of one simulated pedigree (BASELINE configs[4]'s
    model); D from the NumPy restatement of reference scilmm/Matrices/Dominance.py:12-43 (host only: usable without a GPU)."""
    from scilmm_amd.harness.pedigree import simulate_pedigree, ibd_from_parents, drop_unrelated, dominance_from_parents
    par, sex, _ = simulate_pedigree(n, sf, seed)
    A = ibd_from_parents(par)
    D = dominance_from_parents(par, A)
    A2, has, sex2 = drop_unrelated(A, sex)
    D2 = D[has][:, has].tocsr()
    D2.sort_indices()
    return A2, D2, sex2


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
