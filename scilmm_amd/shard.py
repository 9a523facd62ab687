"""Multi-GPU sharding of the hot path (level 1).

Connected components of the relatedness graph are independent diagonal blocks of
V = sum_k s2_k A_k, so factorize / solve / L*R / quadratic forms need NO data-path collective: every rank
owns whole components (flops-weighted bin-packing) and only scalars are all-reduced -- log det V, C'V^-1C,
C'V^-1y and the quadratic forms of the gradient (a few hundred doubles per evaluation, RCCL all-reduce
when the process group's backend is "nccl", gloo in the CPU tests).

World-size invariance: every rank derives its local elimination order from ONE global permutation (the
engine's symbolic analysis of the whole pattern, computed redundantly and deterministically on each
host).  For a block-diagonal matrix the Cholesky factor of a component does not depend on the other
components, so the per-rank factors are exactly the blocks of the single-process factor, and the
simulated vectors P^T L R are fed with the rows of the SAME global normal matrix R (drawn from
``np.random`` identically on every rank, as the reference does at scilmm/SparseCholesky.py:50).  Results
therefore agree across 1/2/4/8 ranks to rounding.

The giant component does not split this way; subtree partition inside it (RCCL extend-add at separator
supernodes) is the next step and is not implemented here.
"""
import numpy as np
import scipy.linalg as la
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components


def component_partition(mats, world):
    """owner[i] = rank that owns individual i.  Components are packed largest-cost-first (LPT)."""
    n = mats[0].shape[0]
    pat = None
    for m in mats:
        b = sp.csr_matrix((np.ones(m.nnz, dtype=np.int8), m.indices, m.indptr), shape=m.shape)
        pat = b if pat is None else pat + b
    ncomp, label = connected_components(pat, directed=False)
    rownnz = np.diff(pat.tocsr().indptr).astype(np.float64)
    cost = np.bincount(label, weights=rownnz ** 2, minlength=ncomp)  # ~ sum of squared column counts
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world)
    comp_owner = np.empty(ncomp, dtype=np.int64)
    for c in order:
        r = int(np.argmin(load))
        comp_owner[c] = r
        load[r] += cost[c]
    return comp_owner[label], load


class ShardedEvaluator(object):
    """One rank's share of the REML evaluation (fused form of SparseCholesky.py:77-117).

    ``make_factor(local_mats, local_perm)`` returns an object with ``factorize(sigma2) -> factor`` where the
    factor obeys the reference protocol plus ``lmul`` -- ``scilmm_amd.factor.Symbolic`` on a GPU, an oracle
    adapter in the CPU tests.  ``quadforms(k, Q)`` must return sum_i (A_k Q)_ic Q_ic on the local rows.
    """

    def __init__(self, mats, C, y, global_perm, rank, world, make_factor, dist=None):
        self.rank, self.world, self.dist = rank, world, dist
        self.n_global = n = mats[0].shape[0]
        owner, self.load = component_partition(mats, world)
        mine = np.where(owner == rank)[0]
        # local order = global elimination order restricted to my individuals
        gpos = np.empty(n, dtype=np.int64)
        gpos[np.asarray(global_perm)] = np.arange(n)
        self.gpos = np.sort(gpos[mine])                    # my positions in the global permuted order
        self.rows = np.asarray(global_perm)[self.gpos]     # my individuals, in elimination order
        local_index = {int(g): i for i, g in enumerate(np.sort(mine))}
        self.sorted_rows = np.sort(mine)
        self.local_perm = np.array([local_index[int(g)] for g in self.rows], dtype=np.int32)
        self.mats = [sp.csr_matrix(m)[self.sorted_rows][:, self.sorted_rows].tocsr() for m in mats]
        for m in self.mats:
            m.sort_indices()
        self.C = np.ascontiguousarray(C[self.sorted_rows])
        self.y = np.ascontiguousarray(y[self.sorted_rows])
        self.engine = make_factor(self.mats, self.local_perm)
        self._fac = None

    def _allreduce(self, vec):
        if self.world == 1 or self.dist is None:
            return vec
        import torch
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def evaluate(self, log_s2, reml=True, sim_num=100):
        s2 = np.exp(np.asarray(log_s2, dtype=float))
        fac = self.engine.factorize(s2)
        n, c = self.n_global, self.C.shape[1]
        nl = self.y.size
        # the SAME global normal matrix on every rank; my rows are my positions in the permuted order
        R = np.random.randn(n, sim_num)[self.gpos]
        # local R rows are in elimination order; the factor's lmul expects them in that order and returns
        # P^T L R in local original order
        Z = fac.lmul(R)
        X = fac(np.hstack([self.C, self.y[:, None], Z]))
        ViC, Viy0, U = X[:, :c], X[:, c], X[:, c + 1:]
        red = self._allreduce(np.concatenate([(self.C.T @ ViC).ravel(), self.C.T @ Viy0, [fac.logdet()]]))
        CtViC = red[:c * c].reshape(c, c)
        G = la.cho_factor(CtViC)
        beta = la.cho_solve(G, red[c * c:c * c + c])
        logdet = red[-1]
        Viy = Viy0 - ViC @ beta
        resid = self.y - self.C @ beta
        pairs = [(a, b) for a in range(c) for b in range(a + 1, c)] if reml else []
        cols = [U, Viy[:, None]]
        if reml:
            cols.append(ViC)
            cols += [(ViC[:, a] + ViC[:, b])[:, None] for a, b in pairs]
        Q = np.ascontiguousarray(np.hstack(cols))
        K = len(self.mats)
        q = np.stack([self.engine.quadforms(k, Q) for k in range(K)]) if nl else np.zeros((K, Q.shape[1]))
        red2 = self._allreduce(np.concatenate([[resid @ Viy], q.ravel()]))
        q = red2[1:].reshape(K, -1)
        nll = 0.5 * (red2[0] + n * np.log(2 * np.pi) + logdet)
        if reml:
            nll += np.log(np.diag(G[0])).sum()
        grad = np.empty(K)
        for k in range(K):
            grad[k] = 0.5 * (q[k, :sim_num].mean() - q[k, sim_num])
            if reml:
                Mk = np.zeros((c, c))
                d = q[k, sim_num + 1: sim_num + 1 + c]
                Mk[np.arange(c), np.arange(c)] = d
                for t, (a, b) in enumerate(pairs):
                    Mk[a, b] = Mk[b, a] = 0.5 * (q[k, sim_num + 1 + c + t] - d[a] - d[b])
                grad[k] -= 0.5 * np.trace(la.cho_solve(G, Mk))
        return nll, grad * s2


def hip_factory(mats, perm):
    """make_factor for the GPU: scilmm_amd.factor.Symbolic already has factorize() and quadforms()."""
    from .factor import Symbolic

    class _Engine(object):
        def __init__(self):
            self.sym = Symbolic(mats, perm=perm)
            self.fac = None

        def factorize(self, s2):
            if self.fac is None:
                self.fac = self.sym.factorize(s2)
            else:
                self.fac.refactorize(s2)
            return self.fac

        def quadforms(self, k, Q):
            return self.sym.quadforms(k, Q)

    return _Engine()
