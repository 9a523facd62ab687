"""ORACLE / TEST INFRASTRUCTURE ONLY -- CPU stand-in for one rank of the distributed chain factorization.

Runs the SAME distribution rule as the HIP engine (scilmm_amd/dist.py, csrc/engine.hip: the trailing run of
single-front levels is the chain; rank r computes chain panel j iff j % world == r, every finished chain panel is
broadcast from its owner; the prelude is replicated) on top of the CPU supernodal code (oracle/supernodal_cpu.c),
with torch.distributed/gloo for the broadcasts.  tests/test_distributed_cpu.py drives scilmm_amd.dist's
DistributedEvaluator with it (world_size 2) and compares against the single-process oracle.
"""
import numpy as np
import scipy.sparse as sp

from . import oracle as O


class CpuChainEngine(object):
    def __init__(self, mats, rank, world, dist=None, perm=None):
        from scilmm_amd.factor import Symbolic
        from scilmm_amd.dist import chain_levels
        self.mats = [sp.csr_matrix(m) for m in mats]
        self.rank, self.world, self.dist = rank, world, dist
        self.n = n = self.mats[0].shape[0]
        sym = Symbolic(self.mats, perm=perm, upload=False)  # host-side analysis only: no device is touched
        arrays = sym.arrays()
        self.perm = arrays["perm"]
        self.cpu = O.SupernodalCPU(arrays, n)
        self.level = sym.get("sn_level")
        self.l0 = chain_levels(sym.get("level_ptr"))
        self.nlevels = int(self.level.max()) + 1 if self.level.size else 0
        self.pat_colptr = sym.get("pat_colptr")
        self.panels_sent = 0
        self.panels_computed = 0
        self._L = None

    def owner(self, lvl):
        return (lvl - self.l0) % self.world

    def factorize(self, sigma2):
        V = None
        for s2, m in zip(sigma2, self.mats):
            V = s2 * m if V is None else V + s2 * m
        p = self.perm
        Lw = sp.tril(V.tocsr()[p][:, p]).tocsc()
        Lw.sort_indices()
        assert np.array_equal(Lw.indptr, self.pat_colptr)  # pattern slots = CSC order of tril(V[P][:,P])
        self.cpu.assemble(Lw.data)
        a = self.cpu.a
        self._L = None
        for s in range(self.cpu.ns):
            lvl = int(self.level[s])
            chain = lvl >= self.l0 and self.world > 1
            if not chain or self.owner(lvl) == self.rank:
                self.cpu.factorize_range(s, s + 1)
                self.panels_computed += 1
            if chain:
                import torch
                m = int(a["sn_rowptr"][s + 1] - a["sn_rowptr"][s])
                w = int(a["sn_start"][s + 1] - a["sn_start"][s])
                view = torch.from_numpy(self.cpu.Lx[a["sn_loff"][s]:a["sn_loff"][s] + m * w])  # shares memory
                self.dist.broadcast(view, src=self.owner(lvl))
                self.panels_sent += 1
        return self

    def P(self):
        return self.perm

    def logdet(self):
        return self.cpu.logdet()

    def solve_local(self, B):
        return self.cpu.solve(B)

    def lmul_local(self, R):
        if self._L is None:
            self._L = self.cpu.L_csc()
        Z = np.empty_like(R)
        Z[self.perm] = self._L @ R  # (L R)[argsort(P)]: row k of L R belongs to individual perm[k]
        return Z

    def quadforms_local(self, k, Q):
        return O.quadforms(self.mats[k], Q)
