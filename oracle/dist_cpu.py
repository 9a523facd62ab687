"""ORACLE / TEST INFRASTRUCTURE ONLY -- CPU stand-in for one rank of the distributed factorization.

Runs the SAME distribution rule as the HIP engine (scilmm_amd/dist.py, csrc/engine.hip) on top of the CPU supernodal
code (oracle/supernodal_cpu.c), with torch.distributed/gloo for the collectives; the rule itself -- owner of every
tail front, rank-local panel offsets, group size, ring slots -- is READ from the library (``scilmm_dist_layout``), so
the rehearsal and the product cannot drift apart:

* rank-local storage: prelude panels + own tail panels + a ring of G slots (a received panel lives in slot j % G);
* fan-out factorization: the owner factors panel j after its LATE update (sources of its own group and of the group
  before it) and broadcasts it; when source group g is complete every rank applies it as one BATCH to all its own
  targets at least two groups ahead;
* forward sweep: the owner pushes x_f into a private accumulator; the ranks all-reduce a block's rows when it is due
  and every rank solves the block (replicated diagonal block); backward: the owner solves and broadcasts x_f;
* L * R: every panel multiplied where it lives (the prelude on rank 0), one all-reduce.

tests/test_distributed_cpu.py drives scilmm_amd.dist's DistributedEvaluator with it (world 2 / 4 / 8) and compares
against the single-process oracle; the counters below let the tests assert who computed / received / stored what.
"""
import numpy as np
import scipy.linalg as la
import scipy.sparse as sp

from . import oracle as O


class CpuChainEngine(object):
    device_resident = False

    def __init__(self, mats, rank, world, dist=None, perm=None):
        from scilmm_amd.factor import Symbolic
        from scilmm_amd.dist import tail_layout
        self.mats = [sp.csr_matrix(m) for m in mats]
        self.rank, self.world, self.dist = rank, world, dist
        self.n = n = self.mats[0].shape[0]
        sym = Symbolic(self.mats, perm=perm, upload=False)  # host-side analysis only: no device is touched
        arrays = sym.arrays()
        self.perm = arrays["perm"]
        ns = len(arrays["sn_start"]) - 1
        self.owner, loff, (self.first, self.Wg, self.G) = tail_layout(sym._h, ns, rank, world)
        self.global_doubles = int(arrays["sn_loff"][-1])
        self.local_doubles = int(loff[-1])
        # assembly map in rank-local offsets; entries of other ranks' tail panels are dropped
        pat_colptr = sym.get("pat_colptr")
        col_front = np.repeat(np.arange(ns), np.diff(arrays["sn_start"]))
        slot_front = np.repeat(col_front, np.diff(pat_colptr))
        keep = (self.owner < 0) | (self.owner == rank)
        delta = loff[:-1] - arrays["sn_loff"][:-1]
        self._asm_keep = keep[slot_front]
        self._asm_dst = (arrays["asm_dst"] + delta[slot_front])[self._asm_keep]
        self._diag_front_keep = keep[col_front]
        self._diag_dst = (arrays["diag_dst"] + delta[col_front])[self._diag_front_keep]
        self._diag_dst_all = arrays["diag_dst"] + delta[col_front]  # (ring offsets for foreign panels: only valid on arrival)
        arrays = dict(arrays)
        arrays["sn_loff"] = np.ascontiguousarray(loff)
        self.cpu = O.SupernodalCPU(arrays, n)
        self.cpu.Lx = np.zeros(self.local_doubles)
        self.keep = keep
        self.level = sym.get("sn_level")
        self.pat_colptr = pat_colptr
        self.ns = ns
        self.nT = ns - self.first
        self.panels_computed = 0   # tail panels this rank factored
        self.panels_received = 0   # tail panels that came in through the ring
        self.batches = 0
        self.collectives = 0
        self._dblk = {}

    # ---- collectives (in place on NumPy memory through torch views)
    def _bcast(self, arr, src):
        import torch
        self.dist.broadcast(torch.from_numpy(arr), src=src)
        self.collectives += 1

    def _allreduce(self, arr, op=None):
        import torch
        if op is None:
            self.dist.all_reduce(torch.from_numpy(arr))
        else:
            self.dist.all_reduce(torch.from_numpy(arr), op=op)
        self.collectives += 1

    def _panel(self, s):
        a = self.cpu.a
        m = int(a["sn_rowptr"][s + 1] - a["sn_rowptr"][s])
        w = int(a["sn_start"][s + 1] - a["sn_start"][s])
        return self.cpu.Lx[a["sn_loff"][s]:a["sn_loff"][s] + m * w], m, w

    def factorize(self, sigma2):
        V = None
        for s2, m in zip(sigma2, self.mats):
            V = s2 * m if V is None else V + s2 * m
        p = self.perm
        Lw = sp.tril(V.tocsr()[p][:, p]).tocsc()
        Lw.sort_indices()
        assert np.array_equal(Lw.indptr, self.pat_colptr)  # pattern slots = CSC order of tril(V[P][:,P])
        self.cpu.Lx[:] = 0.0
        self.cpu.Lx[self._asm_dst] = Lw.data[self._asm_keep]
        first, Wg, nT, world = self.first, self.Wg, self.nT, self.world
        bad = np.array([np.inf])  # first non-positive pivot seen by this rank
        for s in range(self.ns):
            if s < first or world == 1:
                try:
                    self.cpu.factorize_range(s, s + 1)   # replicated prelude
                except O.NotPositiveDefinite as e:
                    bad[0] = min(bad[0], float(e.args[0]))
                    break
                continue
            jj = s - first
            g = jj // Wg
            own = jj % world == self.rank
            if own:
                # late update: prelude descendants + the tail sources of the group before and of the own group so far
                self.cpu.update_from(s, 0, first)
                self.cpu.update_from(s, first + max(0, (g - 1) * Wg), s)
                try:
                    self.cpu.finish(s)
                except O.NotPositiveDefinite as e:
                    bad[0] = min(bad[0], float(e.args[0]))
                    # keep taking part in the collectives: the panel travels as it is, the status is agreed on below
                self.panels_computed += 1
            else:
                self.panels_received += 1
            view, m, w = self._panel(s)  # own storage on the owner, ring slot jj % G elsewhere
            self._bcast(view, jj % world)
            # the diagonal block stays on every rank (the HIP engine broadcasts the inverse diagonal blocks and log-sums
            # with the panel and keeps them replicated): the sweeps solve every block everywhere
            self._dblk[s] = np.tril(view.reshape(w, m)[:, :w].T).copy()
            if (jj + 1) % Wg == 0 or jj == nT - 1:
                # source group g complete: its batch to every own target at least two groups ahead
                for t in range(first + (g + 2) * Wg, self.ns):
                    if (t - first) % world == self.rank:
                        self.cpu.update_from(t, first + g * Wg, first + (g + 1) * Wg)
                self.batches += 1
        if world > 1:
            import torch.distributed as tdist
            self._allreduce(bad, op=tdist.ReduceOp.MIN)
        if np.isfinite(bad[0]):
            raise O.NotPositiveDefinite(int(bad[0]))
        col_front = np.repeat(np.arange(self.ns), np.diff(self.cpu.a["sn_start"]))
        pre = col_front < first if world > 1 else np.ones(col_front.size, dtype=bool)
        ld = np.log(self.cpu.Lx[self._diag_dst_all[pre]]).sum()
        if world > 1:
            ld += sum(np.log(np.diag(self._dblk[s])).sum() for s in range(first, self.ns))
        self._logdet = 2.0 * float(ld)
        return self

    def P(self):
        return self.perm

    def logdet(self):
        return self._logdet

    def solve(self, B):
        """V^-1 B on every rank; B (n, r) in original row order."""
        B = np.asarray(B, dtype=np.float64)
        Y = np.asfortranarray(B.reshape(self.n, -1)[self.perm])
        first, world = self.first, self.world
        if world == 1:
            self.cpu.solve_permuted(Y)
        else:
            ACC = np.zeros_like(Y)
            a = self.cpu.a
            for s in range(self.ns):
                c0, c1 = int(a["sn_start"][s]), int(a["sn_start"][s + 1])
                if s < first:
                    self.cpu.fwd_front(s, Y, Y, 0)
                    self.cpu.fwd_front(s, Y, Y, 1)   # replicated: pushes go straight into the right-hand side
                    continue
                blk = np.ascontiguousarray(ACC[c0:c1])
                self._allreduce(blk)                 # the block's contributions from every owner of an earlier panel
                Y[c0:c1] += blk
                # every rank solves the block itself (replicated diagonal blocks)
                Y[c0:c1] = la.solve_triangular(self._dblk[s], Y[c0:c1], lower=True)
                if (s - first) % world == self.rank:
                    self.cpu.fwd_front(s, Y, ACC, 1)
            for s in range(self.ns - 1, -1, -1):
                c0, c1 = int(a["sn_start"][s]), int(a["sn_start"][s + 1])
                if s < first:
                    self.cpu.bwd_front(s, Y)
                    continue
                if (s - first) % world == self.rank:
                    self.cpu.bwd_front(s, Y)   # needs the whole panel: the owner's job
                blk = np.ascontiguousarray(Y[c0:c1])
                self._bcast(blk, (s - first) % world)
                Y[c0:c1] = blk
        X = np.empty((self.n, Y.shape[1]))
        X[self.perm] = Y
        return X.reshape(B.shape)

    def lmul(self, R):
        """(L R)[argsort(P)] on every rank."""
        R = np.asarray(R, dtype=np.float64)
        Rf = np.asfortranarray(R.reshape(self.n, -1))
        Y = np.zeros_like(Rf)
        for s in range(self.ns):
            if s >= self.first and self.world > 1:
                if (s - self.first) % self.world != self.rank:
                    continue
            elif self.world > 1 and self.rank != 0:
                continue
            self.cpu.lmul_front(s, Rf, Y)
        if self.world > 1:
            Yc = np.ascontiguousarray(Y)
            self._allreduce(Yc)
            Y = Yc
        Z = np.empty((self.n, Rf.shape[1]))
        Z[self.perm] = Y
        return Z.reshape(R.shape)

    def quadforms(self, k, Q):
        return O.quadforms(self.mats[k], Q)

    def spmm(self, k, X):
        """A_k X (the residual of DistributedEvaluator's refinement sweeps)."""
        return self.mats[k] @ np.asarray(X)
