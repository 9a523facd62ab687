"""Multi-GPU factorization of ONE cohort (level 2): one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU / one-GPU rehearsals).

What is distributed, and why this way.  A pedigree factor is a sparse prelude (hundreds of thousands of small
fronts) below a chain of separator supernodes that ends in the dense trailing clique: 76 % of the factor flops at
the 100k config, > 99.9 % at the 1M config (170k-wide clique, 1.6 PFLOP) -- ``profiles/r2_ordering.json`` shows that
neither minimum degree nor nested dissection finds a smaller top separator.  Independent subtrees therefore carry
(almost) none of the work; the separator chain itself has to be shared:

* **chain panels, 1-D block-cyclic.**  The trailing run of single-front levels of the block elimination tree is the
  distributed chain.  Rank r computes chain panel j iff ``j % world == r`` -- left-looking: all the update
  contributions (the extend-add of every descendant, prelude and chain alike) are summed where the panel lives --
  then the finished panel, its inverse diagonal block and its log-sum are **broadcast** to all ranks.  A rank
  applies the contributions of everything older than its previous own panel ahead of time (look-ahead depth =
  world), so only the last ``world`` source panels sit on the critical path.
* **the prelude is replicated** (every rank factors it; < 0.1 % of the flops at 1M).  Its contribution blocks are
  consumed locally by whoever owns the target chain panel: no reduce is needed for them.
* **after the factorization every rank holds the complete factor**, so the solves need no sweep across GPUs:
  the right-hand-side COLUMNS are split over the ranks (103 columns per REML evaluation) and all-gathered.

The communication pattern (broadcast of panel j from ``j % world``, in chain order; all-gather of solution
columns) lives here, in Python, on top of ``torch.distributed``; the HIP engine only orders its streams around the
callbacks (``scilmm_dist_init``).  The same evaluator drives a CPU engine in ``tests/test_distributed_cpu.py``
(world_size 2, gloo) so that the N > 1 logic is covered without a GPU.
"""
import ctypes as C

import numpy as np
import scipy.linalg as la

from . import _lib


def chain_levels(level_ptr):
    """First level of the distributed chain: the trailing run of levels that hold exactly one front."""
    nlev = len(level_ptr) - 1
    l0 = nlev
    while l0 > 0 and level_ptr[l0] - level_ptr[l0 - 1] == 1:
        l0 -= 1
    return l0


def column_chunks(r, world):
    """Contiguous column ranges [c0, c1) per rank (the first r % world ranks get one column more)."""
    base, extra = divmod(r, world)
    out, c = [], 0
    for k in range(world):
        w = base + (1 if k < extra else 0)
        out.append((c, c + w))
        c += w
    return out


class ColumnSplit(object):
    """Runs a column-wise linear map (solve, L*R, quadratic forms) on this rank's share of the columns and
    all-gathers the result, so that every rank ends with the full array."""

    def __init__(self, rank, world, dist=None, device=None):
        self.rank, self.world, self.dist, self.device = rank, world, dist, device

    def _gather(self, local, shapes):
        if self.world == 1 or self.dist is None:
            return [local]
        import torch
        dev = self.device if self.device is not None else "cpu"
        width = max(s[1] - s[0] for s in shapes)
        pad = np.zeros(local.shape[:-1] + (width,))
        pad[..., :local.shape[-1]] = local
        mine = torch.from_numpy(np.ascontiguousarray(pad)).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return [p.cpu().numpy()[..., :s[1] - s[0]] for p, s in zip(parts, shapes)]

    def apply(self, fn, B):
        """fn maps an (n, k) block to an (n, k) block column by column; B is (n, r)."""
        B = np.asarray(B, dtype=np.float64)
        chunks = column_chunks(B.shape[1], self.world)
        c0, c1 = chunks[self.rank]
        local = fn(np.ascontiguousarray(B[:, c0:c1])) if c1 > c0 else np.zeros((B.shape[0], 0))
        return np.concatenate(self._gather(local, chunks), axis=1)

    def apply_reduce(self, fn, B):
        """fn maps an (n, k) block to k numbers (one per column)."""
        B = np.asarray(B, dtype=np.float64)
        chunks = column_chunks(B.shape[1], self.world)
        c0, c1 = chunks[self.rank]
        local = np.asarray(fn(np.ascontiguousarray(B[:, c0:c1])), dtype=np.float64) if c1 > c0 else np.zeros(0)
        return np.concatenate(self._gather(local, chunks))


class HipChainEngine(object):
    """The HIP engine as one rank of a distributed factorization: torch-owned factor storage, broadcasts issued by
    the engine's callback on a dedicated torch stream."""

    def __init__(self, mats, rank, world, dist, device, perm=None, ordering="amd"):
        import torch
        from .factor import Symbolic
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._comm_stream = torch.cuda.Stream(device=self.device)
        self._bufs = [None, None, None]
        self._cb_error = None

        def _comm(ctx, op, buffer, offset, count, root):
            try:
                t = self._bufs[buffer][offset:offset + count]
                with torch.cuda.stream(self._comm_stream):
                    if op == 0:
                        dist.broadcast(t, src=root)
                    else:
                        dist.all_reduce(t)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self._cb_error = e
                return -1

        self._cb = _lib.COMM_FN(_comm)  # keep the callback object alive as long as the engine
        self.sym = Symbolic(mats, perm=perm, ordering=ordering, upload=False)
        _lib.check(_lib.lib().scilmm_dist_init(self.sym._h, rank, world, C.c_void_p(self._comm_stream.cuda_stream),
                                               self._cb, None), self.sym._h)
        self.sym.upload_values()
        nL, nI, nS = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().scilmm_factor_sizes(self.sym._h, C.byref(nL), C.byref(nI), C.byref(nS)), self.sym._h)
        self._bufs = [torch.empty(nL.value, dtype=torch.float64, device=self.device),
                      torch.empty(nI.value, dtype=torch.float64, device=self.device),
                      torch.zeros(nS.value, dtype=torch.float64, device=self.device)]
        torch.cuda.synchronize(self.device)
        h = C.c_void_p()
        _lib.check(_lib.lib().scilmm_factor_create_external(self.sym._h, C.c_void_p(self._bufs[0].data_ptr()),
                                                            C.c_void_p(self._bufs[1].data_ptr()),
                                                            C.c_void_p(self._bufs[2].data_ptr()), C.byref(h)), self.sym._h)
        from .factor import Factor
        self.fac = Factor.__new__(Factor)
        self.fac.sym, self.fac.n, self.fac._h, self.fac._s2 = self.sym, self.sym.n, h, None
        self.n = self.sym.n

    def factorize(self, sigma2):
        try:
            self.fac.refactorize(sigma2)
        except Exception:
            if self._cb_error is not None:
                raise self._cb_error
            raise
        return self

    def P(self):
        return self.sym.P()

    def logdet(self):
        return self.fac.logdet()

    def solve_local(self, B):
        return self.fac(B)

    def lmul_local(self, R):
        return self.fac.lmul(R)

    def quadforms_local(self, k, Q):
        return self.sym.quadforms(k, Q)


class DistributedEvaluator(object):
    """One REML likelihood + gradient evaluation (fused form of reference SparseCholesky.py:77-117) with the factor
    distributed over the ranks of ``dist``.  ``engine``: ``HipChainEngine`` on GPUs, a CPU stand-in in the tests.
    Every rank returns the same (nll, grad): the column results are all-gathered, the scalars are computed
    redundantly from them."""

    def __init__(self, engine, mats, C_cov, y, rank, world, dist=None, device=None):
        self.engine, self.mats, self.C, self.y = engine, mats, np.asarray(C_cov, float), np.asarray(y, float)
        self.split = ColumnSplit(rank, world, dist, device)

    def evaluate(self, log_s2, reml=True, sim_num=100):
        s2 = np.exp(np.asarray(log_s2, dtype=float))
        eng = self.engine
        eng.factorize(s2)
        n, c = self.y.size, self.C.shape[1]
        R = np.random.randn(n, sim_num)  # the same global normal matrix on every rank (SparseCholesky.py:50)
        Z = self.split.apply(eng.lmul_local, R)
        X = self.split.apply(eng.solve_local, np.hstack([self.C, self.y[:, None], Z]))
        ViC, Viy0, U = X[:, :c], X[:, c], X[:, c + 1:]
        G = la.cho_factor(self.C.T @ ViC)
        beta = la.cho_solve(G, self.C.T @ Viy0)
        Viy = Viy0 - ViC @ beta
        resid = self.y - self.C @ beta
        nll = 0.5 * (resid @ Viy + n * np.log(2 * np.pi) + eng.logdet())
        if reml:
            nll += np.log(np.diag(G[0])).sum()
        pairs = [(a, b) for a in range(c) for b in range(a + 1, c)] if reml else []
        cols = [U, Viy[:, None]]
        if reml:
            cols.append(ViC)
            cols += [(ViC[:, a] + ViC[:, b])[:, None] for a, b in pairs]
        Q = np.ascontiguousarray(np.hstack(cols))
        K = len(self.mats)
        grad = np.empty(K)
        for k in range(K):
            q = self.split.apply_reduce(lambda blk, k=k: eng.quadforms_local(k, blk), Q)
            grad[k] = 0.5 * (q[:sim_num].mean() - q[sim_num])
            if reml:
                Mk = np.zeros((c, c))
                d = q[sim_num + 1: sim_num + 1 + c]
                Mk[np.arange(c), np.arange(c)] = d
                for t, (a, b) in enumerate(pairs):
                    Mk[a, b] = Mk[b, a] = 0.5 * (q[sim_num + 1 + c + t] - d[a] - d[b])
                grad[k] -= 0.5 * np.trace(la.cho_solve(G, Mk))
        return nll, grad * s2
