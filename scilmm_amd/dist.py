"""Multi-GPU factorization of ONE cohort (level 2): one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU / one-GPU rehearsals).

What is distributed, and why this way.  A pedigree factor is a sparse prelude (hundreds of thousands of small
fronts) below the dense tail: the separator supernodes that end in the trailing clique -- 76 % of the factor flops at
the 100k config, > 99.9 % at the 1M config (178k columns, 1.6 PFLOP, 123 GB) -- and ``profiles/r2_ordering.json`` shows
that neither minimum degree nor nested dissection finds a smaller top separator.  Independent subtrees carry (almost)
none of the work, and the tail alone outgrows one GPU's memory at BASELINE configs[4] (3M individuals: ~1.3 TB).  So:

* **tail panels, 1-D block-cyclic, FAN-OUT, rank-local storage.**  Tail front ``dense_first + j`` belongs to rank
  ``j % world``.  A rank stores the prelude, its own tail panels and a ring of a few panel slots.  A finished panel is
  **broadcast** by its owner into the receivers' ring; every rank applies it to its own later targets and drops it:
  per rank ``nnz(L_tail) / world`` + ring instead of the whole factor.  To keep the accumulators of the update kernel
  in registers over a deep K, sources are applied a GROUP (8 panels) at a time: when group g is complete, one batch
  launch updates every own target at least two groups ahead; the sources of a target's own group and of the one
  before it are its "late" update, right before its diagonal block is factored.
* **the prelude is replicated** (every rank factors it; < 0.1 % of the flops at 1M).
* **the triangular sweeps** run where the panels are: forward, the owner pushes x_f through its panel into a private
  accumulator and the ranks all-reduce the 128 rows of a block when it is due (the inverse diagonal blocks are
  replicated, so every rank then solves the block itself); backward, the owner solves its block and broadcasts x_f;
  L*R is multiplied panel by panel where the panels live and summed by one all-reduce.  Every rank ends with the full
  result, like the single-GPU calls.

The communication pattern lives here, in Python, on top of ``torch.distributed``; the HIP engine only orders its
streams around the callbacks (``scilmm_dist_init``).  The same evaluator drives a CPU engine with the same rule in
``tests/test_distributed_cpu.py`` (world_size 2 / 4 / 8, gloo) so that the N > 1 logic is covered without a GPU.
"""
import ctypes as C

import numpy as np
import scipy.linalg as la

from . import _lib


def column_chunks(r, world):
    """Contiguous column ranges [c0, c1) per rank (the first r % world ranks get one column more)."""
    base, extra = divmod(r, world)
    out, c = [], 0
    for k in range(world):
        w = base + (1 if k < extra else 0)
        out.append((c, c + w))
        c += w
    return out


def tail_layout(sym_handle, nsuper, rank, world):
    """The distribution rule as data, straight from the library (``scilmm_dist_layout``): owner per front (-1 =
    replicated), rank-local panel offsets (last entry: doubles of local panel storage), (first distributed front,
    group size, ring slots).  Host only."""
    owner = np.empty(nsuper, dtype=np.int32)
    loff = np.empty(nsuper + 1, dtype=np.int64)
    params = np.empty(3, dtype=np.int32)
    _lib.check(_lib.lib().scilmm_dist_layout(sym_handle, rank, world, _lib.ptr(owner), _lib.ptr(loff), _lib.ptr(params)), sym_handle)
    return owner, loff, (int(params[0]), int(params[1]), int(params[2]))


class HipChainEngine(object):
    """The HIP engine as one rank of a distributed factorization: torch-owned storage (this rank's share of the factor,
    the sweeps' work buffer), collectives issued by the engine's callback on a dedicated torch stream."""

    device_resident = True

    def __init__(self, mats, rank, world, dist, device, perm=None, ordering="amd", cache=None):
        import torch
        from .factor import Symbolic
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._comm_stream = torch.cuda.Stream(device=self.device)
        self._bufs = [None, None, None, None]
        self._cb_error = None
        self.collectives = 0

        def _comm(ctx, op, buffer, offset, count, root):
            try:
                t = self._bufs[buffer][offset:offset + count]
                with torch.cuda.stream(self._comm_stream):
                    if op == 0:
                        dist.broadcast(t, src=root)
                    elif op == 1:
                        dist.all_reduce(t)
                    else:
                        dist.all_reduce(t, op=dist.ReduceOp.MIN)
                self.collectives += 1
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self._cb_error = e
                return -1

        self._cb = _lib.COMM_FN(_comm)  # keep the callback object alive as long as the engine
        # (cache: a directory -- /dev/shm on one node -- where the first rank to analyse the pattern leaves the image of the
        #  analysis for the others: scilmm_symbolic_save / _load)
        if cache and dist is not None and world > 1:
            # ONE writer per cache: rank 0 analyses the pattern and publishes the image, the others load it after the barrier
            # (ranks missing the cache together would each analyse and each write the same image: correct -- the library
            # writes under a private name and renames -- but world x the work and the memory)
            if rank == 0:
                self.sym = Symbolic(mats, perm=perm, ordering=ordering, upload=False, cache=cache)
            dist.barrier()
            if rank != 0:
                self.sym = Symbolic(mats, perm=perm, ordering=ordering, upload=False, cache=cache)
        else:
            self.sym = Symbolic(mats, perm=perm, ordering=ordering, upload=False, cache=cache)
        L = _lib.lib()
        _lib.check(L.scilmm_dist_init(self.sym._h, rank, world, C.c_void_p(self._comm_stream.cuda_stream), self._cb, None), self.sym._h)
        self.sym.upload_values()
        nL, nI, nS, nW = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(L.scilmm_factor_sizes(self.sym._h, C.byref(nL), C.byref(nI), C.byref(nS)), self.sym._h)
        _lib.check(L.scilmm_dist_work_size(self.sym._h, C.byref(nW)), self.sym._h)
        self.local_factor_doubles = nL.value
        self._bufs = [torch.empty(nL.value, dtype=torch.float64, device=self.device),
                      torch.empty(nI.value, dtype=torch.float64, device=self.device),
                      torch.zeros(nS.value, dtype=torch.float64, device=self.device),
                      torch.zeros(nW.value, dtype=torch.float64, device=self.device)]
        torch.cuda.synchronize(self.device)
        _lib.check(L.scilmm_dist_set_work(self.sym._h, C.c_void_p(self._bufs[3].data_ptr())), self.sym._h)
        h = C.c_void_p()
        _lib.check(L.scilmm_factor_create_external(self.sym._h, C.c_void_p(self._bufs[0].data_ptr()),
                                                   C.c_void_p(self._bufs[1].data_ptr()),
                                                   C.c_void_p(self._bufs[2].data_ptr()), C.byref(h)), self.sym._h)
        from .factor import Factor
        self.fac = Factor.from_handle(self.sym, h)
        self.n = self.sym.n

    def _guard(self, fn, *a):
        try:
            return fn(*a)
        except Exception:
            if self._cb_error is not None:
                err, self._cb_error = self._cb_error, None
                raise err
            raise

    def factorize(self, sigma2):
        """Every rank raises NotPositiveDefiniteError together (the status word is all-reduced inside the library)."""
        self._guard(self.fac.refactorize, sigma2)
        return self

    def P(self):
        return self.sym.P()

    def logdet(self):
        return self.fac.logdet()

    # host arrays in, host arrays out (every rank gets the full result)
    def solve(self, B):
        return self._guard(self.fac, B)

    def lmul(self, R):
        return self._guard(self.fac.lmul, R)

    def quadforms(self, k, Q):
        return self.sym.quadforms(k, Q)

    # device tensors in, device tensors out: the n x 100 blocks of an evaluation never leave HBM
    def solve_t(self, dB):
        dX = self.torch.empty_like(dB)
        self.torch.cuda.synchronize(self.device)
        self._guard(self.fac.solve_dev, C.c_void_p(dB.data_ptr()), dB.shape[1], C.c_void_p(dX.data_ptr()))
        self._guard(self.sym.sync)
        return dX

    def lmul_t(self, dR):
        dZ = self.torch.empty_like(dR)
        self.torch.cuda.synchronize(self.device)
        self._guard(self.fac.lmul_dev, C.c_void_p(dR.data_ptr()), dR.shape[1], C.c_void_p(dZ.data_ptr()))
        self._guard(self.sym.sync)
        return dZ

    @property
    def front_bits(self):
        return getattr(self.sym, "front_bits", 64)

    def spmm_t(self, k, dX):
        """A_k X for a device block (n x r, any r): the residual of a refinement sweep (scilmm_spmm_dev)."""
        dY = self.torch.empty_like(dX)
        self.torch.cuda.synchronize(self.device)
        self.sym.spmm_dev(k, C.c_void_p(dX.data_ptr()), dX.shape[1], C.c_void_p(dY.data_ptr()))
        self.sym.sync()
        return dY

    def quadforms_t(self, k, dQ):
        dq = self.torch.empty(dQ.shape[1], dtype=self.torch.float64, device=self.device)
        self.torch.cuda.synchronize(self.device)
        self.sym.quadforms_dev(k, C.c_void_p(dQ.data_ptr()), dQ.shape[1], C.c_void_p(dq.data_ptr()))
        self.sym.sync()
        return dq


class DistributedEvaluator(object):
    """One REML likelihood + gradient evaluation (fused form of reference SparseCholesky.py:77-117) with the factor
    distributed over the ranks of ``dist``.  ``engine``: ``HipChainEngine`` on GPUs (the n x 100 blocks stay in HBM as
    torch tensors; only c + 1 solution columns and the quadratic forms reach the host), a CPU stand-in in the tests
    (NumPy arrays).  The factor sweeps are collective (every rank takes part in every column); the quadratic forms --
    a streaming pass over the replicated A_k values -- are split by COLUMNS over the ranks and all-gathered.
    Every rank returns the same (nll, grad)."""

    REFINE_STEPS = 2  # (as scilmm_amd.factor.Factor.REFINE_STEPS: each sweep gains ~7 digits on a factor with fp32-product fronts)

    def __init__(self, engine, mats, C_cov, y, rank, world, dist=None, device=None, refine_steps=None):
        """``refine_steps``: iterative-refinement sweeps of the fused solve against the exact V = sum_k sigma2_k A_k.  None =
        what the factor needs: ``REFINE_STEPS`` when the engine runs fp32-product fronts (``front_bits == 32``, BASELINE
        configs[4]), none on an fp64 factor."""
        self.engine, self.mats, self.C, self.y = engine, mats, np.asarray(C_cov, float), np.asarray(y, float)
        self.rank, self.world, self.dist, self.device = rank, world, dist, device
        self.refine_steps = refine_steps
        self.last_refinement = []   # max |correction| / max |x| of every sweep of the last evaluation (diagnostics, tests)

    def _apply_V(self, s2, X, on_dev):
        """V X = sum_k sigma2_k A_k X for an n x r block that every rank holds, with the COLUMNS split over the ranks (a
        streaming pass over the replicated A_k values per column, as for the quadratic forms) and the column blocks put
        together again by one all-reduce of the zero-padded block -- the residual of a refinement sweep, the counterpart
        of ``_finish_on_device``'s K x ``spmm_dev`` (SparseCholesky.py) on the multi-rank path."""
        eng = self.engine
        r = X.shape[1]
        q0, q1 = column_chunks(r, self.world)[self.rank]
        if on_dev:
            torch = eng.torch
            out = torch.zeros_like(X)
            if q1 > q0:
                mine = X[:, q0:q1].contiguous()
                acc = None
                for k in range(len(self.mats)):
                    t = eng.spmm_t(k, mine)
                    acc = t.mul_(float(s2[k])) if acc is None else acc.add_(t, alpha=float(s2[k]))
                out[:, q0:q1] = acc
            if self.dist is not None:   # (also at world 1: the one-rank RCCL smoke test runs the collective for real)
                torch.cuda.synchronize(eng.device)
                self.dist.all_reduce(out)
                torch.cuda.synchronize(eng.device)
            return out
        out = np.zeros_like(X)
        if q1 > q0:
            mine = np.ascontiguousarray(X[:, q0:q1])
            for k in range(len(self.mats)):
                out[:, q0:q1] += float(s2[k]) * np.asarray(eng.spmm(k, mine))
        if self.dist is not None:
            import torch
            self.dist.all_reduce(torch.from_numpy(out))
        return out

    def _refined_solve(self, s2, B, on_dev):
        """X = V^-1 B through the collective sweeps, followed by ``refine_steps`` sweeps x += V^-1 (B - V x)."""
        eng = self.engine
        X = eng.solve_t(B) if on_dev else eng.solve(B)
        steps = self.refine_steps
        if steps is None:
            steps = self.REFINE_STEPS if getattr(eng, "front_bits", 64) == 32 else 0
        self.last_refinement = []
        for _ in range(steps):
            res = B - self._apply_V(s2, X, on_dev)
            dx = eng.solve_t(res) if on_dev else eng.solve(res)
            self.last_refinement.append(float(abs(dx).max() / abs(X).max()))
            X = X + dx
        return X

    def _gather_vec(self, local, chunks):
        """local: this rank's slice of a length-r vector (host array) -> the full vector on every rank."""
        if self.dist is None:
            return np.asarray(local)
        import torch
        width = max(b - a for a, b in chunks)
        pad = np.zeros(width)
        pad[:len(local)] = local
        dev = self.device if self.device is not None else "cpu"
        mine = torch.from_numpy(pad).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return np.concatenate([p.cpu().numpy()[:b - a] for p, (a, b) in zip(parts, chunks)])

    def evaluate(self, log_s2, reml=True, sim_num=100):
        s2 = np.exp(np.asarray(log_s2, dtype=float))
        eng = self.engine
        eng.factorize(s2)
        n, c = self.y.size, self.C.shape[1]
        R = np.random.randn(n, sim_num)  # the same global normal matrix on every rank (SparseCholesky.py:50)
        on_dev = getattr(eng, "device_resident", False)
        if on_dev:
            torch = eng.torch
            dZ = eng.lmul_t(torch.from_numpy(R).to(eng.device))
            head = torch.from_numpy(np.ascontiguousarray(np.hstack([self.C, self.y[:, None]]))).to(eng.device)
            dX = self._refined_solve(s2, torch.cat([head, dZ], dim=1).contiguous(), True)
            del dZ
            Xh = dX[:, :c + 1].cpu().numpy()
            ViC, Viy0 = Xh[:, :c], Xh[:, c]
        else:
            Z = eng.lmul(R)
            X = self._refined_solve(s2, np.hstack([self.C, self.y[:, None], Z]), False)
            ViC, Viy0, U = X[:, :c], X[:, c], X[:, c + 1:]
        G = la.cho_factor(self.C.T @ ViC)
        beta = la.cho_solve(G, self.C.T @ Viy0)
        Viy = Viy0 - ViC @ beta
        resid = self.y - self.C @ beta
        nll = 0.5 * (resid @ Viy + n * np.log(2 * np.pi) + eng.logdet())
        if reml:
            nll += np.log(np.diag(G[0])).sum()
        pairs = [(a, b) for a in range(c) for b in range(a + 1, c)] if reml else []
        tail = [Viy[:, None]]
        if reml:
            tail.append(ViC)
            tail += [(ViC[:, a] + ViC[:, b])[:, None] for a, b in pairs]
        tail = np.ascontiguousarray(np.hstack(tail))
        rq = sim_num + tail.shape[1]
        chunks = column_chunks(rq, self.world)
        q0, q1 = chunks[self.rank]
        if on_dev:
            dQ = torch.cat([dX[:, c + 1:], torch.from_numpy(tail).to(eng.device)], dim=1)[:, q0:q1].contiguous()
            del dX
        else:
            Q = np.ascontiguousarray(np.hstack([U, tail])[:, q0:q1])
        K = len(self.mats)
        grad = np.empty(K)
        for k in range(K):
            if q1 > q0:
                loc = eng.quadforms_t(k, dQ).cpu().numpy() if on_dev else np.asarray(eng.quadforms(k, Q))
            else:
                loc = np.zeros(0)
            q = self._gather_vec(loc, chunks)
            grad[k] = 0.5 * (q[:sim_num].mean() - q[sim_num])
            if reml:
                Mk = np.zeros((c, c))
                d = q[sim_num + 1: sim_num + 1 + c]
                Mk[np.arange(c), np.arange(c)] = d
                for t, (a, b) in enumerate(pairs):
                    Mk[a, b] = Mk[b, a] = 0.5 * (q[sim_num + 1 + c + t] - d[a] - d[b])
                grad[k] -= 0.5 * np.trace(la.cho_solve(G, Mk))
        return nll, grad * s2
