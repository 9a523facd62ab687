"""Drop-in surface of reference ``scilmm/SparseCholesky.py`` on top of the MI355X engine.

Same module-level names, argument order and return values as the reference (file:line cited per
function), so that ``from scilmm_amd import SparseCholesky, REML, HE, run_estimates`` and the
``--A --phe --cov --reml --ignore_indices`` CLI behave like the reference's.  What differs is HOW one
likelihood evaluation is computed when ``cholesky_func`` is the HIP ``SparseCholesky``:

* the pattern of V is analysed ONCE per list of matrices (the reference re-runs CHOLMOD's analysis on
  every evaluation, SparseCholesky.py:22-26,92) and the A_k values live in HBM;
* V = sum s2_k A_k is assembled on the device (no scipy ``csr*scalar``/``+=``/``tocsc``, :55-59);
* the four solves of an evaluation (:30,:32,:52,:100) become ONE multi-column sweep over
  ``[C | y | P^T L R]`` (V^-1(y - C b) = V^-1 y - (V^-1 C) b by linearity);
* ``factor.L().dot(R)[argsort(P)]`` (:50-51) is a supernodal product on the device (no CSC export);
* ``sum((A_k U) * U)`` (:65), ``v' A_k v`` (:66) and ``C' V^-1 A_k V^-1 C`` (:70) come from one fused
  SpMM+reduce per matrix (off-diagonal entries of the c x c block by polarisation).

Any other callable obeying the factor protocol (``factor(b)``, ``.L()``, ``.P()``, ``.logdet()``) is
accepted as ``cholesky_func`` exactly as in the reference and takes the reference-shaped path; there is
no silent fallback: constructing ``SparseCholesky()`` without the built HIP library raises.
"""
import os
import time

import numpy as np
import pandas as pd
import scipy.linalg as la
from scipy.linalg import blas as _blas
import scipy.optimize as optimize
import scipy.sparse as sparse
from scipy.io import mmread

from . import _lib
from .factor import Symbolic

np.set_printoptions(precision=3, linewidth=200)
pd.set_option('display.width', 200)


def _pattern_key(mats):
    key = []
    for m in mats:
        key.append((m.shape[0], m.nnz, hash(m.indptr.tobytes()), hash(m.indices.tobytes())))
    return tuple(key)


class SparseCholesky(object):
    """Reference ``SparseCholesky`` (SparseCholesky.py:16-26): a callable ``V -> factor``.

    ``use_long`` / ``mode`` are accepted for signature compatibility (the engine is always supernodal with 64-bit
    offsets).  ``ordering_method``: ``'nesdis'`` (what the reference passes, SparseCholesky.py:17) runs the engine's
    nested dissection, ``'amd'`` its approximate minimum degree, ``'best'`` both and keeps the one with fewer factor
    flops, ``'natural'`` none; a ``perm=`` array is honoured exactly.  The constructor default is ``'default'`` =
    minimum degree rather than the reference's ``'nesdis'``: on every pedigree configuration of the ordering study
    (``profiles/r2_ordering.json``) minimum degree gives the smaller factor (nested dissection: +8 % flops at 100k,
    x2 at 30k), and the analysis runs once per pattern either way.
    """

    def __init__(self, use_long=False, mode='supernodal', ordering_method='default', perm=None, fused=True,
                 exact_trace=False, cache_dir=None, metrics=None, front_bits=64):
        _lib.lib()  # fail loudly when the HIP library is not built
        self._use_long = use_long
        self._mode = mode
        self._ordering_method = ordering_method
        self._perm = perm
        self.fused = fused
        # opt-in: tr(V^-1 A_k) of the gradient computed exactly -- selected inverse on the supernodal
        # factor, on the device -- instead of by the reference's Monte-Carlo estimate (SparseCholesky.py:49-52, :65): see
        # _exact_traces.  Default off: the default behaviour has to be the reference's stochastic estimator.
        self.exact_trace = exact_trace
        # cache_dir: keep the image of the symbolic analysis there (once per pattern instead of once per process; the
        # reference writes its stage artefacts to files as well, scilmm/IBDCompute.py:82-84); also SCILMM_SYMBOLIC_CACHE.
        self.cache_dir = cache_dir
        # metrics: path of a JSON-lines file (also SCILMM_METRICS) that receives one record per likelihood evaluation --
        # sigma2, nll, gradient, and the device timers of that evaluation
        self.metrics = metrics or os.environ.get("SCILMM_METRICS")
        # front_bits=32 (opt-in; BASELINE configs[4]'s "fp64 factor with fp32 MFMA fronts"): the products of the dense-tail
        # update run on the fp32 matrix pipe (scilmm_set_front_precision), everything else stays fp64 and the factor's solves
        # are refined against the exact V (Factor.__call__); 1.4 - 1.5x the fp64 speed at the 300k / 1M configs.  A pattern
        # whose dense tail is narrower than the engine's threshold keeps the fp64 path.
        if front_bits not in (32, 64):
            raise ValueError("front_bits must be 32 or 64")
        self.front_bits = front_bits
        self._n_eval = 0
        self._cache = {}

    def _ordering(self):
        # 'nesdis' (the reference's choice, SparseCholesky.py:17) = the engine's nested dissection; 'default' and
        # 'amd' = its approximate minimum degree; 'best' = whichever of the two gives fewer factor flops
        return {'natural': 'natural', 'nesdis': 'nesdis', 'best': 'best'}.get(self._ordering_method, 'amd')

    @staticmethod
    def _quick_id(mats):
        """Identity of a list of CSR matrices that an in-place edit cannot slip past: object ids and buffer
        addresses (the cache keeps strong references, so ids are not recycled) plus a FULL-pass checksum of the
        values (two streaming BLAS reductions, ~0.2 s per 1e9 entries)."""
        out = []
        for m in mats:
            d = m.data
            # two level-1 BLAS reductions (multithreaded: ~10 ms per 1e8 entries where ndarray.sum took 50): the sum of
            # squares, and the sum of products of neighbours, which -- unlike the first -- also moves when an entry
            # changes its sign or two entries swap
            if d.size and d.dtype == np.float64 and d.flags.c_contiguous:
                c1 = float(_blas.ddot(d, d))
                c2 = float(_blas.ddot(d[:-1], d[1:])) if d.size > 1 else 0.0
            else:
                c1, c2 = (float(np.dot(d, d)), float(d.sum())) if d.size else (0.0, 0.0)
            out.append((id(m), m.shape, m.nnz, d.ctypes.data if d.size else 0, m.indices.ctypes.data if d.size else 0,
                        m.indptr.ctypes.data, c1, c2))
        return tuple(out)

    def engine_for(self, mats):
        """Symbolic analysis (cached per sparsity pattern) with the values of ``mats`` resident in HBM."""
        if all(sparse.isspmatrix_csr(m) for m in mats):
            qid = self._quick_id(mats)
            if getattr(self, "_last_qid", None) == qid:
                return self._last_sym  # same objects as in the previous evaluation: nothing to re-hash
        else:
            qid = None
        held = list(mats)  # strong references: the ids inside qid stay valid while it is cached
        # (a matrix with unsorted indices is copied before sorting: the caller's arrays are never touched)
        mats = [m if (sparse.isspmatrix_csr(m) and m.has_sorted_indices) else sparse.csr_matrix(m).sorted_indices()
                for m in mats]
        key = _pattern_key(mats)
        hit = self._cache.get(key)
        if hit is None:
            sym = Symbolic(mats, perm=self._perm, ordering=self._ordering(), cache=self.cache_dir)
            if self.front_bits == 32:
                try:
                    sym.set_front_precision(32)
                except _lib.ScilmmError:
                    pass  # (no dense tail wide enough: this pattern is factorized in fp64 throughout)
            self._cache = {key: (sym, [m.data.copy() for m in mats])}
        else:
            sym, datas = hit
            for k, m in enumerate(mats):
                if not np.array_equal(datas[k], m.data):
                    sym.set_values(k, m.data)
                    datas[k] = m.data.copy()
        self._last_qid, self._last_sym, self._last_held = qid, sym, held
        return sym

    def release_factors(self):
        """Free the numeric factor(s) kept between evaluations (the symbolic analysis and the A_k values stay)."""
        self.__dict__.get('_factor_state', {}).clear()

    def __call__(self, sparse_mat):
        sym = self.engine_for([sparse_mat])
        return sym.factorize([1.0])


def _is_hip(cholesky_func):
    return isinstance(cholesky_func, SparseCholesky)


# ---------------------------------------------------------------------------------------------------
# Reference-shaped building blocks (used with any factor-protocol object)

def estimate_fixed_effects(factor, y, covariates):
    """GLS fixed effects (SparseCholesky.py:29-34)."""
    invV_C = factor(covariates)
    L_CT_invV_C = la.cho_factor(covariates.T.dot(invV_C))
    fixed_effects = la.cho_solve(L_CT_invV_C, covariates.T.dot(factor(y)))
    mu = covariates.dot(fixed_effects)
    return invV_C, L_CT_invV_C, mu, fixed_effects


def negative_log_likelihood(factor, y, invV_y, mu, L_CT_invV_C, reml):
    """-log-likelihood up to constants (SparseCholesky.py:37-46)."""
    n = y.size
    nll = 0.5 * ((y - mu).dot(invV_y) + n * np.log(2 * np.pi) + factor.logdet())
    if reml:
        nll += np.sum(np.log(np.diag(L_CT_invV_C[0])))
    return nll


def simulate_vector(factor, n, sim_num, p_inv):
    """U = V^-1 (P^T L R) with R ~ N(0, I) from the global legacy RNG (SparseCholesky.py:49-52)."""
    R = np.random.randn(n, sim_num)
    if hasattr(factor, 'lmul'):
        return factor(factor.lmul(R))
    return factor(factor.L().dot(R)[p_inv])


def matrices_weighted_sum(mats, sig2g_array):
    """V = sum_k s2_k A_k as CSC (SparseCholesky.py:55-59); only used on the reference-shaped path."""
    V = sig2g_array[0] * mats[0]
    for i in range(1, len(sig2g_array)):
        V = V + sig2g_array[i] * mats[i]
    return V.tocsc()


def compute_gradients(sig2g_array, mats, sim_vec, invV_y, reml, invV_C, L_CT_invV_C):
    """Monte-Carlo gradient of the negative log-likelihood (SparseCholesky.py:62-74)."""
    grad = np.zeros(len(sig2g_array))
    for k in range(len(sig2g_array)):
        AU = mats[k].dot(sim_vec)
        tr_est = np.mean(np.sum(AU * sim_vec, axis=0))
        quad = invV_y.dot(mats[k].dot(invV_y))
        grad[k] = 0.5 * (tr_est - quad)
        if reml:
            grad[k] -= 0.5 * np.trace(la.cho_solve(L_CT_invV_C, invV_C.T.dot(mats[k].dot(invV_C))))
    return grad


def _polar_columns(c):
    pairs = [(a, b) for a in range(c) for b in range(a + 1, c)]
    return pairs


def _device_buffers():
    """torch (device memory and copies only -- plumbing) if it can reach the GPU the engine runs on, else None: the
    evaluation then goes through host buffers, same arithmetic, more PCIe traffic."""
    if os.environ.get("SCILMM_HOST_BUFFERS") == "1" or os.environ.get("SCILMM_NO_TORCH") == "1":
        return None
    try:
        import torch
        return torch if torch.cuda.is_available() else None
    except Exception:
        return None


def _solve_on_device(torch, sym, fac, dB, sig2g_array=None):
    """X = V^-1 B for a device block (n x r torch tensor) on the resident factor, without leaving HBM.  On a factor with
    fp32-product fronts the solve is refined against the exact V = sum_k sigma2_k A_k (what ``Factor.__call__`` does through
    host buffers): each sweep = K SpMMs + one more solve and gains ~7 digits."""
    import ctypes
    vp = ctypes.c_void_p
    r_all = dB.shape[1]
    dX = torch.empty_like(dB)
    torch.cuda.synchronize()
    fac.solve_dev(vp(dB.data_ptr()), r_all, vp(dX.data_ptr()))
    sym.sync()
    if getattr(sym, "front_bits", 64) == 32:
        s2 = fac._s2 if sig2g_array is None else np.asarray(sig2g_array, dtype=float)
        if s2 is None:
            raise _lib.ScilmmError("refinement needs the sigma2 the resident factor was built from")
        dY, dRes = torch.empty_like(dB), torch.empty_like(dB)
        for _ in range(fac.REFINE_STEPS):
            dRes.copy_(dB)
            for k in range(len(s2)):
                torch.cuda.synchronize()
                sym.spmm_dev(k, vp(dX.data_ptr()), r_all, vp(dY.data_ptr()))
                sym.sync()
                dRes.sub_(dY, alpha=float(s2[k]))
            torch.cuda.synchronize()
            fac.solve_dev(vp(dRes.data_ptr()), r_all, vp(dY.data_ptr()))
            sym.sync()
            dX.add_(dY)
        del dY, dRes
        torch.cuda.synchronize()
    return dX


def _spmm_on_device(torch, sym, k, dX):
    """A_k X for a device block (``scilmm_spmm_dev``)."""
    import ctypes
    dX = dX.contiguous()
    dY = torch.empty_like(dX)
    torch.cuda.synchronize()
    sym.spmm_dev(k, ctypes.c_void_p(dX.data_ptr()), dX.shape[1], ctypes.c_void_p(dY.data_ptr()))
    sym.sync()
    return dY


def _finish_on_device(torch, sym, fac, R, sig2g_array, covariates, y, reml, sim_num):
    """Everything after the factorization of one evaluation with the n x 100 blocks kept in HBM: R goes down once,
    Z = P^T L R, the 103-column solve and the K fused SpMM + reduce calls read and write device buffers through the
    `_dev` entry points, and only the c + 1 solution columns the host algebra needs (and the quadratic forms) come
    back.  Same operations in the same order as the host-buffer path (fused form of SparseCholesky.py:88-109)."""
    import ctypes
    n, c = y.size, covariates.shape[1]
    vp = ctypes.c_void_p
    dR = torch.from_numpy(np.ascontiguousarray(R)).cuda()
    dZ = torch.empty_like(dR)
    torch.cuda.synchronize()
    fac.lmul_dev(vp(dR.data_ptr()), sim_num, vp(dZ.data_ptr()))
    sym.sync()
    del dR
    head = torch.from_numpy(np.ascontiguousarray(np.hstack([covariates, y[:, None]]))).cuda()
    dB = torch.cat([head, dZ], dim=1).contiguous()
    del dZ
    # (fp32-product fronts: refined against the exact V on the device, see _solve_on_device)
    dX = _solve_on_device(torch, sym, fac, dB, sig2g_array)
    del dB
    Xh = dX[:, :c + 1].cpu().numpy()
    invV_C, invV_y0 = Xh[:, :c], Xh[:, c]
    L_CT_invV_C = la.cho_factor(covariates.T.dot(invV_C))
    beta = la.cho_solve(L_CT_invV_C, covariates.T.dot(invV_y0))
    mu = covariates.dot(beta)
    invV_y = invV_y0 - invV_C.dot(beta)
    nll = negative_log_likelihood(fac, y, invV_y, mu, L_CT_invV_C, reml)
    tail = [invV_y[:, None]]
    pairs = _polar_columns(c) if reml else []
    if reml:
        tail.append(invV_C)
        for a, b in pairs:
            tail.append((invV_C[:, a] + invV_C[:, b])[:, None])
    dQ = torch.cat([dX[:, c + 1:], torch.from_numpy(np.ascontiguousarray(np.hstack(tail))).cuda()], dim=1).contiguous()
    del dX
    rq = dQ.shape[1]
    dq = torch.empty(rq, dtype=torch.float64, device=dQ.device)
    grad = np.zeros(len(sig2g_array))
    for k in range(len(sig2g_array)):
        torch.cuda.synchronize()
        sym.quadforms_dev(k, vp(dQ.data_ptr()), rq, vp(dq.data_ptr()))
        sym.sync()
        q = dq.cpu().numpy()
        grad[k] = 0.5 * (np.mean(q[:sim_num]) - q[sim_num])
        if reml:
            M = np.zeros((c, c))
            d = q[sim_num + 1: sim_num + 1 + c]
            M[np.arange(c), np.arange(c)] = d
            for t, (a, b) in enumerate(pairs):
                M[a, b] = M[b, a] = 0.5 * (q[sim_num + 1 + c + t] - d[a] - d[b])
            grad[k] -= 0.5 * np.trace(la.cho_solve(L_CT_invV_C, M))
    return nll, grad, fac


EXACT_TRACE_BLOCK = 512


def _exact_traces(fac, mats):
    """tr(V^-1 A_k) for every k, exactly: the SELECTED INVERSE on the supernodal factor -- Takahashi
    recursion on the device, in place (``scilmm_selected_inverse``: twice the factorization's flops, nothing crosses
    PCIe, no size limit) -- and one streaming pass over each A_k's pattern (``scilmm_inverse_traces``).  Removes the
    random vectors, and with them np.random, from the objective.  CONSUMES the factor: every caller is done with its
    solves by then and refactorizes at the next evaluation."""
    return fac.inverse_traces()


def _exact_traces_bruteforce(fac, mats):
    """The round-2 form, kept as an independent check of the selected inverse (tests): V^-1 E_b for blocks E_b of identity
    columns (one multi-right-hand-side sweep of the device factor per block), contracted with the matching columns of
    A_k -- n / EXACT_TRACE_BLOCK sweeps and n^2 doubles over PCIe."""
    n = fac.n
    cscs = [sparse.csc_matrix(m) for m in mats]
    tr = np.zeros(len(mats))
    for b0 in range(0, n, EXACT_TRACE_BLOCK):
        b1 = min(n, b0 + EXACT_TRACE_BLOCK)
        E = np.zeros((n, b1 - b0))
        E[np.arange(b0, b1), np.arange(b1 - b0)] = 1.0
        X = fac(E)
        for k, m in enumerate(cscs):
            blk = m[:, b0:b1]
            tr[k] += float(blk.multiply(X).sum())
    return tr


def _evaluate_hip(sig2g_array, cholesky_func, mats, covariates, y, reml, sim_num):
    """One likelihood + gradient evaluation on the device (fused form of SparseCholesky.py:88-109)."""
    sym = cholesky_func.engine_for(mats)
    state = cholesky_func.__dict__.setdefault('_factor_state', {})
    fac = state.get(id(sym))
    n = y.size
    c = covariates.shape[1]
    exact = bool(getattr(cholesky_func, 'exact_trace', False))
    if exact:
        # no random vectors: factorize, solve for [C | y], exact traces
        if fac is None:
            state.clear()
            fac = state[id(sym)] = sym.factorize(sig2g_array)
        else:
            fac.refactorize(sig2g_array)
        sim_num = 0
        Z = np.zeros((n, 0))
    elif fac is None:
        state.clear()
        fac = state[id(sym)] = sym.factorize(sig2g_array)
        R = np.random.randn(n, sim_num)
    else:
        # the device factorizes while the host draws the normal matrix (same np.random stream as the reference:
        # one randn(n, sim_num) per evaluation, SparseCholesky.py:50); wait() raises NotPositiveDefiniteError
        fac.refactorize_async(sig2g_array)
        R = np.random.randn(n, sim_num)
        fac.wait()
    dev = None if exact or not cholesky_func.fused else _device_buffers()
    if dev is not None:
        return _finish_on_device(dev, sym, fac, R, sig2g_array, covariates, y, reml, sim_num)
    if not exact:
        Z = fac.lmul(R)
    if cholesky_func.fused:
        X = fac(np.hstack([covariates, y[:, None], Z]))
        invV_C, invV_y0, U = X[:, :c], X[:, c], X[:, c + 1:]
        L_CT_invV_C = la.cho_factor(covariates.T.dot(invV_C))
        beta = la.cho_solve(L_CT_invV_C, covariates.T.dot(invV_y0))
        mu = covariates.dot(beta)
        invV_y = invV_y0 - invV_C.dot(beta)
    else:
        invV_C, L_CT_invV_C, mu, beta = estimate_fixed_effects(fac, y, covariates)
        invV_y = fac(y - mu)
        U = fac(Z) if Z.shape[1] else Z
    nll = negative_log_likelihood(fac, y, invV_y, mu, L_CT_invV_C, reml)
    # gradient: one fused SpMM+reduce per matrix over [U | v | C-columns and pairwise sums]
    cols = [U, invV_y[:, None]]
    pairs = _polar_columns(c) if reml else []
    if reml:
        cols.append(invV_C)
        for a, b in pairs:
            cols.append((invV_C[:, a] + invV_C[:, b])[:, None])
    Q = np.ascontiguousarray(np.hstack(cols))
    grad = np.zeros(len(sig2g_array))
    if exact and getattr(cholesky_func, '_before_traces', None) is not None:
        # (the selected inverse CONSUMES the factor: whatever else needs solves at this sigma2 -- the AI matrix of _ai_reml --
        # runs here, between the evaluation's own sweep and the inverse)
        cholesky_func._hook_result = cholesky_func._before_traces(fac)
    traces = _exact_traces(fac, mats) if exact else None
    for k in range(len(sig2g_array)):
        q = sym.quadforms(k, Q)
        grad[k] = 0.5 * ((traces[k] if exact else np.mean(q[:sim_num])) - q[sim_num])
        if reml:
            M = np.zeros((c, c))
            d = q[sim_num + 1: sim_num + 1 + c]
            M[np.arange(c), np.arange(c)] = d
            for t, (a, b) in enumerate(pairs):
                M[a, b] = M[b, a] = 0.5 * (q[sim_num + 1 + c + t] - d[a] - d[b])
            grad[k] -= 0.5 * np.trace(la.cho_solve(L_CT_invV_C, M))
    return nll, grad, fac


def _write_metrics(cholesky_func, mats, sig2g_array, nll, grad, seconds, reml, sim_num):
    """One JSON line per likelihood evaluation: what was evaluated, what came out, where the time went
    (HIP-event timers of this evaluation's assembly / factorization / sweeps / L*R / quadratic forms)."""
    import json
    sym = cholesky_func.engine_for(mats)
    info = sym.info()
    cholesky_func._n_eval += 1
    rec = {"evaluation": cholesky_func._n_eval, "time": time.time(), "n": int(info.n), "K": int(info.K), "nnzL": int(info.nnzL),
           "factor_flops": info.flops, "reml": bool(reml), "sim_num": int(sim_num), "sigma2": np.asarray(sig2g_array).tolist(),
           "nll": float(nll), "grad_sigma2": np.asarray(grad).tolist(), "seconds": seconds,
           "device_ms": {k: v for k, v in sym.timing().items() if k.endswith("_ms")},
           "launches": int(sym.timing()["n_launches"]), "symbolic_from_cache": bool(getattr(sym, "from_cache", False))}
    with open(cholesky_func.metrics, "a") as fh:
        fh.write(json.dumps(rec) + "\n")


def bolt_gradient_estimation(log_sig2g_array, cholesky_func, mats, covariates, y, reml, sim_num, verbose,
                             take_exp=True):
    """nll and d nll / d log(sigma2) at one point (SparseCholesky.py:77-117)."""
    sig2g_array = np.exp(log_sig2g_array) if take_exp else np.asarray(log_sig2g_array, dtype=float)
    if verbose:
        t0 = time.time()
        print('estimating nll and its gradient at:', sig2g_array)
    if _is_hip(cholesky_func):
        t_eval = time.time()
        nll, grad, _ = _evaluate_hip(sig2g_array, cholesky_func, mats, covariates, y, reml, sim_num)
        if cholesky_func.metrics:
            _write_metrics(cholesky_func, mats, sig2g_array, nll, grad, time.time() - t_eval, reml, sim_num)
    else:
        V = matrices_weighted_sum(mats, sig2g_array)
        n = V.shape[0]
        factor = cholesky_func(V)
        P_inv = np.argsort(factor.P())
        invV_C, L_CT_invV_C, mu, _ = estimate_fixed_effects(factor, y, covariates)
        invV_y = factor(y - mu)
        nll = negative_log_likelihood(factor, y, invV_y, mu, L_CT_invV_C, reml)
        sim_vec = simulate_vector(factor, n, sim_num, P_inv)
        grad = compute_gradients(sig2g_array, mats, sim_vec, invV_y, reml, invV_C, L_CT_invV_C)
    if take_exp:
        grad = grad * sig2g_array
    if verbose:
        print("grad : ", grad)
        print('nll: %0.8e   computation time: %0.2f seconds' % (nll, time.time() - t0))
    return nll, grad


def estimate_var_comps(cholesky_func, mats, covariates, y, reml=True, sim_num=100, verbose=True, aireml=False):
    """HE start, then L-BFGS-B on log sigma2 (SparseCholesky.py:120-144)."""
    engine = None
    if _is_hip(cholesky_func):
        # the analysis is needed for the first evaluation anyway; with the values in HBM the HE moments are two
        # streaming passes on the device instead of SciPy sparse arithmetic on the host
        engine = (cholesky_func.engine_for(mats), list(range(len(mats) - 1)))
    he_est = HE(mats[:-1], covariates, y, compute_stderr=False, engine=engine)
    x0 = np.concatenate((he_est, [1 - he_est.sum()]))
    if np.any(x0 < 0):
        x0 = np.ones((len(mats)))
    x0 = x0 / x0.sum()
    if aireml:
        s2 = _ai_reml(cholesky_func, mats, covariates, y, x0, reml, sim_num, verbose)
        if not (_is_hip(cholesky_func) and getattr(cholesky_func, 'exact_trace', False)):
            return s2
        # exact traces make the objective deterministic, so the end point can be CHECKED: the average-information step can
        # stall next to a boundary the likelihood is flat along (a component pulled towards zero iteration after iteration).
        # Stationary = no component can still lower the likelihood by the reference's tolerance (ftol 1e-7, relative): raising a
        # component with a negative derivative by its own size (at least 0.1 % of the total variance), or lowering one with a
        # positive derivative to zero.  Only a point that fails this is handed to the reference's optimiser to finish.
        nll_end, g_end = _ai_reml.last
        reach = np.where(g_end < 0, np.maximum(s2, 1e-3 * s2.sum()), s2)
        if np.max(np.abs(g_end) * reach) <= 1e-7 * max(abs(nll_end), 1.0):
            return s2
        x0 = np.maximum(s2, 1e-8 * s2.sum())
    optObj = optimize.minimize(bolt_gradient_estimation, np.log(x0),
                               args=(cholesky_func, mats, covariates, y, reml, sim_num, verbose, True),
                               jac=True, method='L-BFGS-B', options={'eps': 1e-5, 'ftol': 1e-7})
    if not optObj.success:
        print('optimization failed with message: %s' % optObj.message)
    return np.exp(optObj.x)


def _ai_reml(cholesky_func, mats, covariates, y, x0, reml, sim_num, verbose, max_iter=50, tol=1e-8, ftol=1e-7):
    """Average-information REML: the branch the reference leaves as a stub (``raise NotImplementedError('AI-REML is
    broken')``, SparseCholesky.py:128-130).  Newton-type iteration on sigma2 itself,
        sigma2 <- sigma2 - AI^-1 g,      g_k  = d nll / d sigma2_k      (the estimator of compute_gradients, :62-74),
                                         AI_kl = 1/2 y' P A_k P A_l P y  (average of observed and expected information),
    with P y = V^-1 (y - C beta).  The gradient comes from ONE call of ``bolt_gradient_estimation`` (take_exp=False): the
    same factorization, fused sweep and Monte-Carlo -- or, with ``SparseCholesky(exact_trace=True)``, exact -- trace as the
    L-BFGS-B path; the AI matrix costs K more single-column solves on the factor that evaluation left resident.  Steps are
    halved until the likelihood does not rise (or the Newton decrement falls); a component a step would drive through zero is
    pulled to a tenth of its value instead.  The Monte-Carlo trace uses COMMON
    random numbers: every evaluation of the fit restarts np.random from the state it had on entry (the probe vectors
    are the same at every sigma2, as in BOLT-REML), so the objective is one smooth function and the iteration converges
    like a Newton method instead of wandering inside the estimator's noise; on return the stream has advanced by one
    evaluation.  Stops on the reference's L-BFGS-B tolerance (relative likelihood change <= 1e-7) or a relative step
    below ``tol``.  With exact traces (deterministic objective) ``estimate_var_comps`` hands the end point to the reference's L-BFGS-B, which
    returns at once from a stationary point and finishes the job where the iteration stalled next to a boundary.
    Opt-in: ``REML(..., aireml=True)``; the default optimiser stays the reference's L-BFGS-B."""
    s2 = np.asarray(x0, dtype=float).copy()
    rng_state = np.random.get_state()

    def information(fac):
        # P y and P A_k P y through the resident factor: 1 + K multi-column solves
        hip = _hip_factor_of(fac, mats)
        if hip is not None:
            # ... in HBM: V^-1 [C | y] (one sweep), K SpMMs on one column, one K-column sweep
            torch, sym = hip
            pr = _DeviceProjector(torch, sym, fac, covariates, y)
            Py_d = pr.project_solved(pr.Viy[:, None])
            APy_d = torch.cat([_spmm_on_device(torch, sym, k, Py_d) for k in range(len(mats))], dim=1)
            AI_d = 0.5 * (APy_d.T @ pr.project(APy_d)).cpu().numpy()
            return 0.5 * (AI_d + AI_d.T)
        ViC = fac(covariates)
        G = la.cho_factor(covariates.T.dot(ViC))

        def proj(Z):
            ViZ = fac(Z)
            return ViZ - ViC.dot(la.cho_solve(G, covariates.T.dot(ViZ)))

        Py = proj(y)
        APy = np.stack([m.dot(Py) for m in mats], axis=1)          # n x K
        PAPy = proj(APy)
        AI = 0.5 * APy.T.dot(PAPy)
        return 0.5 * (AI + AI.T)

    exact_hip = _is_hip(cholesky_func) and bool(getattr(cholesky_func, 'exact_trace', False))

    def evaluate(v):
        np.random.set_state(rng_state)
        if exact_hip:
            # exact traces: the selected inverse overwrites the factor at the end of the evaluation, so the information
            # matrix is computed INSIDE it, right before the inverse (_evaluate_hip's hook)
            cholesky_func._before_traces = information
            try:
                nll, grad = bolt_gradient_estimation(v, cholesky_func, mats, covariates, y, reml, sim_num, False, take_exp=False)
            finally:
                cholesky_func._before_traces = None
            return nll, grad, cholesky_func._hook_result
        nll, grad = bolt_gradient_estimation(v, cholesky_func, mats, covariates, y, reml, sim_num, False, take_exp=False)
        if _is_hip(cholesky_func):
            fac = cholesky_func._factor_state[id(cholesky_func.engine_for(mats))]
        else:
            fac = cholesky_func(matrices_weighted_sum(mats, v))
        return nll, grad, information(fac)

    def decrement(g, M):
        # Newton decrement g' AI^-1 g: the size of the gradient in the metric of the information matrix
        try:
            return float(g.dot(la.solve(M, g, assume_a='pos')))
        except la.LinAlgError:
            return float(g.dot(g / np.maximum(np.diag(M), 1e-300)))

    nll, grad, AI = evaluate(s2)
    for it in range(max_iter):
        try:
            step = -la.solve(AI, grad, assume_a='pos')
        except la.LinAlgError:
            step = -grad / np.maximum(np.diag(AI), 1e-300)
        lam = decrement(grad, AI)
        t = 1.0
        accepted = False
        for halving in range(12):
            # a component the step would drive through zero is pulled to a tenth of its value instead (the other components
            # still take their step: scaling the whole step by the most constrained component stalls the iteration next to a
            # boundary the likelihood is flat along)
            cand = np.maximum(s2 + t * step, 0.1 * s2)
            if np.all(cand > 1e-10 * s2.sum()):
                nll_c, grad_c, AI_c = evaluate(cand)
                # the estimator's gradient is not exactly the derivative of its likelihood value (exact log-det, estimated
                # trace): a step counts as progress when either of them says so
                # (with exact traces the objective is deterministic and its gradient exact: only a likelihood that does not
                # rise counts)
                if nll_c <= nll + 1e-12 * abs(nll) or (not exact_hip and decrement(grad_c, AI_c) < lam):
                    accepted = True
                    break
            t *= 0.5
        if verbose:
            print('AI-REML iteration %d: sigma2 %s nll %.10e step %.3g' % (it, cand if accepted else s2, nll_c if accepted else nll, t))
        if not accepted:
            break
        done = np.abs(cand - s2).max() <= tol * np.abs(s2).max() or abs(nll - nll_c) <= ftol * max(abs(nll), abs(nll_c), 1.0)
        s2, nll, grad, AI = cand, nll_c, grad_c, AI_c
        if done:
            break
    _ai_reml.last = (float(nll), np.asarray(grad, dtype=float).copy())  # (end point's objective and d nll / d sigma2)
    return s2


def _final_factor(cholesky_func, mats, coefficients):
    """The reference factorizes once more at the optimum (SparseCholesky.py:182).  On the device the evaluation
    factor is re-used (refactorize): a second resident factor would double the largest allocation (123 GB at 1M)."""
    if _is_hip(cholesky_func):
        sym = cholesky_func.engine_for(mats)
        fac = cholesky_func.__dict__.setdefault('_factor_state', {}).get(id(sym))
        if fac is not None:
            # (the optimiser usually stops ON its last evaluation: the factor is then already the one asked for)
            return fac if fac.holds(coefficients) else fac.refactorize(coefficients)
        fac = sym.factorize(coefficients)
        cholesky_func._factor_state.clear()
        cholesky_func._factor_state[id(sym)] = fac
        return fac
    return cholesky_func(matrices_weighted_sum(mats, coefficients))


def _hip_factor_of(factor, mats):
    """(torch, symbolic) when ``factor`` is a resident HIP factor of exactly these matrices and device buffers are available
    -- the post-fit algebra then runs as a few multi-column device sweeps -- else None (the reference-shaped host loops)."""
    from .factor import Factor
    if not isinstance(factor, Factor) or getattr(factor, "_s2", None) is None:
        return None
    sym = factor.sym
    if factor._epoch != sym._values_epoch:   # (the resident values changed after this factorization)
        return None
    if len(mats) != sym.K or any(m.shape != (sym.n, sym.n) or m.nnz != sym._indices[k].size for k, m in enumerate(mats)):
        return None
    torch = _device_buffers()
    return None if torch is None else (torch, sym)


class _DeviceProjector(object):
    """P z = V^-1 z - V^-1 C (C' V^-1 C)^-1 C' V^-1 z on device blocks (SparseCholesky.py:153-155), with V^-1 [C | y] from
    ONE multi-column sweep of the resident factor."""

    def __init__(self, torch, sym, factor, covariates, y):
        self.torch, self.sym, self.factor = torch, sym, factor
        self.c = c = covariates.shape[1]
        self.dC = torch.from_numpy(np.ascontiguousarray(covariates, dtype=np.float64)).cuda()
        dB = torch.cat([self.dC, torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)[:, None]).cuda()], dim=1).contiguous()
        dX = _solve_on_device(torch, sym, factor, dB)
        self.ViC, self.Viy = dX[:, :c].contiguous(), dX[:, c].contiguous()
        self.CtViC = (self.dC.T @ self.ViC).cpu().numpy()          # c x c: host LAPACK, like the reference
        self.L_CT_Vinv_C = la.cho_factor(self.CtViC)

    def solve(self, dZ):
        return _solve_on_device(self.torch, self.sym, self.factor, dZ.contiguous())

    def project_solved(self, ViZ):
        """P Z given V^-1 Z."""
        coef = la.cho_solve(self.L_CT_Vinv_C, (self.dC.T @ ViZ).cpu().numpy())
        return ViZ - self.ViC @ self.torch.from_numpy(np.ascontiguousarray(coef)).to(ViZ.device)

    def project(self, dZ):
        return self.project_solved(self.solve(dZ))


def _compute_hess_device(torch, sym, mats, covariates, factor, y, proj=None):
    """compute_hess with every n-vector in HBM: V^-1 [C | y] (one sweep), A_j P y for all j (K SpMMs on one column), their
    projection (one K-column sweep), A_i (P A_j P y) for all i, j (K SpMMs on K columns), their projection (one K^2-column
    sweep) and K^2 dot products with y -- 3 sweeps and 2 K SpMM calls instead of 2 + K + K (K + 1) / 2 single-column sweeps
    and as many host SciPy products with PCIe round trips (SparseCholesky.py:147-168; 56 s of the 239 s 1M fit in round 3)."""
    K = len(mats)
    proj = proj or _DeviceProjector(torch, sym, factor, covariates, y)
    dy = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).cuda()
    Py = proj.project_solved(proj.Viy[:, None])                                               # n x 1
    APy = torch.cat([_spmm_on_device(torch, sym, j, Py) for j in range(K)], dim=1)            # n x K: column j = A_j P y
    PAPy = proj.project(APy)                                                                  # n x K: column j = P A_j P y
    cols = torch.cat([_spmm_on_device(torch, sym, i, PAPy) for i in range(K)], dim=1)         # n x K^2: column i K + j = A_i P A_j P y
    Pcols = proj.project(cols)
    v = (dy @ Pcols).cpu().numpy().reshape(K, K)                                              # v[i, j] = y' P A_i P A_j P y
    hess = np.empty((K, K))
    for j in range(K):
        for i in range(j + 1):
            hess[i, j] = hess[j, i] = -0.5 * v[i, j]
    return hess


def compute_hess(mats, covariates, factor, y):
    """AI-style Hessian of the log-likelihood by projected solves (SparseCholesky.py:147-168).  On a resident HIP factor
    the same quantities are computed by a few multi-column device sweeps (``_compute_hess_device``)."""
    hip = _hip_factor_of(factor, mats)
    if hip is not None:
        return _compute_hess_device(hip[0], hip[1], mats, covariates, factor, y)
    K = len(mats)
    Vinv_C = factor(covariates)
    L_CT_Vinv_C = la.cho_factor(covariates.T.dot(Vinv_C))

    def project(z):
        Vinv_z = factor(z)
        return Vinv_z - Vinv_C.dot(la.cho_solve(L_CT_Vinv_C, covariates.T.dot(Vinv_z)))

    Py = project(y)
    hess = np.empty((K, K))
    for j in range(K):
        P_Hj_Py = project(mats[j].dot(Py))
        for i in range(j + 1):
            hess[i, j] = hess[j, i] = -0.5 * y.dot(project(mats[i].dot(P_Hj_Py)))
    return hess


def compute_varcomp_stderr(mats, covariates, factor, y, sim_num):
    """sqrt(diag((-H)^-1) (1 + 1/s)) (SparseCholesky.py:171-174)."""
    hess = compute_hess(mats, covariates, factor, y)
    return np.sqrt(np.diag(la.inv(-hess)) * (1 + 1.0 / sim_num))


def REML(cholesky_func, mats, covariates, y, reml=True, sim_num=100, verbose=False, aireml=False):
    """REML fit (SparseCholesky.py:177-189). Returns the reference's dict of three arrays.  ``aireml=True`` (not in the
    reference's signature, whose AI-REML branch is a stub) selects the average-information iteration instead of L-BFGS-B."""
    y = y / y.std()
    mats = list(mats) + [sparse.eye(y.shape[0]).tocsr()]
    varcomp_estimates = estimate_var_comps(cholesky_func, mats, covariates, y, reml, sim_num, verbose, aireml=aireml)
    factor = _final_factor(cholesky_func, mats, varcomp_estimates)
    hip = _hip_factor_of(factor, mats)
    if hip is not None:
        # everything after the fit from ONE V^-1 [C | y] sweep + the two sweeps of the Hessian, all in HBM
        proj = _DeviceProjector(hip[0], hip[1], factor, covariates, y)
        fixed_effects = la.cho_solve(proj.L_CT_Vinv_C, (proj.dC.T @ proj.Viy).cpu().numpy())
        hess = _compute_hess_device(hip[0], hip[1], mats, covariates, factor, y, proj=proj)
        sigmas_sigmas = np.sqrt(np.diag(la.inv(-hess)) * (1 + 1.0 / sim_num))
        del proj
    else:
        _, _, _, fixed_effects = estimate_fixed_effects(factor, y, covariates)
        sigmas_sigmas = compute_varcomp_stderr(mats, covariates, factor, y, sim_num)
    del factor
    if _is_hip(cholesky_func):
        cholesky_func.release_factors()
    return {"covariance coefficients": varcomp_estimates,
            "covariates coefficients": fixed_effects,
            "covariance std": sigmas_sigmas}


def HE(mat_list, cov, y, MQS=False, verbose=False, sim_num=100, compute_stderr=False, y2=None, engine=None):
    """Haseman-Elston moment estimator (SparseCholesky.py:192-281); REML's starting point (:121).

    ``engine=(symbolic, ks)`` (not in the reference): ``mat_list[i]`` is matrix ``ks[i]`` of a device-resident
    ``scilmm_amd.factor.Symbolic``; the matrix-sized moments -- ``sum(A_i o A_j)`` and ``y'A_i y`` -- are then one
    streaming pass each over values already in HBM (``scilmm_he_moments``, ``scilmm_quadforms``) instead of SciPy
    ``multiply().sum()`` / ``dot`` on the host.  Same estimator, same result to rounding.

    The Monte-Carlo standard error follows the evident intent of the reference loop (:259-278), whose
    inner variable shadowing (``mat_i``/``mat_j``) makes it raise for more than one matrix.
    """
    mat_list = list(mat_list)
    CTC = cov.T.dot(cov)
    y = y - cov.dot(np.linalg.solve(CTC, cov.T.dot(y)))
    y = y / y.std()
    if y2 is not None:
        y2 = y2 - cov.dot(np.linalg.solve(CTC, cov.T.dot(y2)))
        y2 = y2 / y2.std()
        y = np.concatenate((y, y2))
        for m_i, m in enumerate(mat_list):
            z = sparse.csr_matrix((m.shape[0], m.shape[0]))
            mat_list[m_i] = sparse.vstack([sparse.hstack([z, m]), sparse.hstack([m, z])]).tocsr()
    K = len(mat_list)
    n = y.shape[0]
    q = np.zeros(K)
    S = np.zeros((K, K))
    on_device = engine is not None and not MQS and y2 is None and all(sparse.issparse(m) for m in mat_list)
    if on_device:
        sym, ks = engine
        yc = np.ascontiguousarray(y[:, None])
        for i in range(K):
            q[i] = sym.quadforms(ks[i], yc)[0] - mat_list[i].diagonal().dot(y ** 2)
            for j in range(i + 1):
                fro, dg = sym.he_moments(ks[i], ks[j])
                S[i, j] = S[j, i] = fro - dg
    for i, mat_i in enumerate(mat_list if not on_device else []):
        if MQS:
            q[i] = y.dot(mat_i.dot(y)) - y.dot(y)
        elif sparse.issparse(mat_i):
            q[i] = y.dot(mat_i.dot(y)) - mat_i.diagonal().dot(y ** 2)
        else:
            q[i] = y.dot(mat_i.dot(y)) - np.diag(mat_i).dot(y ** 2)
        for j in range(i + 1):
            mat_j = mat_list[j]
            if MQS:
                S[i, j] = (mat_i.multiply(mat_j)).sum() - (n - 1)
            elif sparse.issparse(mat_i):
                S[i, j] = (mat_i.multiply(mat_j)).sum() - mat_i.diagonal().dot(mat_j.diagonal())
            else:
                S[i, j] = np.einsum('ij,ij->', mat_i, mat_j) - np.diag(mat_i).dot(np.diag(mat_j))
            S[j, i] = S[i, j]
    he_est = np.linalg.solve(S, q)
    if not compute_stderr:
        return he_est
    H = mat_list[0] * he_est[0]
    for mat_k, sigma2_k in zip(mat_list[1:], he_est[1:]):
        H = H + mat_k * sigma2_k
    H = H + sparse.eye(n, format='csr') * (1.0 - he_est.sum())
    V_q = np.empty((K, K))
    torch = _device_buffers() if (sim_num is not None and all(sparse.issparse(m) for m in mat_list)) else None
    if torch is not None:
        # the Monte-Carlo variance (:259-278) with the n x sim_num blocks in HBM: every product is one pass of
        # scilmm_csr_spmm_dev over a device-resident copy of the matrix (no symbolic analysis: HE never factorizes); the
        # normal blocks come from the same np.random stream, pair by pair, as on the host
        dmats = [_lib.DeviceCSR(m, torch) for m in mat_list]
        rest = 1.0 - float(he_est.sum())

        def H_dot(t):
            out = t * rest
            for dm, s2k in zip(dmats, he_est):
                out.add_(dm.dot(t), alpha=float(s2k))
            return out

        for i in range(K):
            for j in range(i + 1):
                d_sim = torch.from_numpy(np.random.randn(n, sim_num)).cuda()
                t = dmats[j].dot(d_sim).sub_(d_sim)
                t = H_dot(t)
                t = dmats[i].dot(t).sub_(t)
                t = H_dot(t)
                V_q[i, j] = V_q[j, i] = 2.0 * float((d_sim * t).sum(dim=0).mean())
        del dmats
    for i, mat_i in enumerate(mat_list if torch is None else []):
        for j, mat_j in enumerate(mat_list[:i + 1]):
            if sim_num is None:
                HAi = H.dot(mat_i) - H
                HAj = H.dot(mat_j) - H
                V_q[i, j] = 2 * (HAi.multiply(HAj)).sum()
            else:
                sim_y = np.random.randn(n, sim_num)
                t = mat_j.dot(sim_y) - sim_y
                t = H.dot(t)
                t = mat_i.dot(t) - t
                t = H.dot(t)
                V_q[i, j] = 2 * np.mean(np.einsum('ij,ij->j', sim_y, t))
            V_q[j, i] = V_q[i, j]
    var_he_est = np.linalg.solve(S, np.linalg.solve(S, V_q).T).T
    return he_est, np.sqrt(np.diag(var_he_est))


def run_estimates(A, df_phe, df_cov, reml=False, ignore_indices=False, df_phe2=None):
    """Align inputs, drop unrelated individuals, standardise covariates, fit (SparseCholesky.py:350-395)."""
    A = sparse.csr_matrix(A)
    if not ignore_indices:
        indices = sorted(set(df_cov.index) & set(df_phe.index) & set(A.indices))
        df_cov = df_cov.loc[indices]
        df_phe = df_phe.loc[indices]
        if df_phe2 is not None:
            df_phe2 = df_phe2.loc[indices]
        A = A[indices][:, indices]
    has_relatives = np.asarray(A.sum(axis=1))[:, 0] > 1
    if any(~has_relatives):
        A = A[has_relatives][:, has_relatives]
        df_cov = df_cov.loc[has_relatives]
        df_phe = df_phe.loc[has_relatives]
        if df_phe2 is not None:
            df_phe2 = df_phe2.loc[has_relatives]
    A.eliminate_zeros()
    y = np.asarray(df_phe.values, dtype=float).reshape(-1)
    y2 = None
    if df_phe2 is not None:
        y2 = np.asarray(df_phe2.values, dtype=float).reshape(-1)
        assert not reml
    if isinstance(df_cov, pd.Series):
        df_cov = df_cov.to_frame()
    df_cov = df_cov.copy()
    df_cov['intercept'] = 1
    cov = df_cov.values.copy().astype(float)
    cov[:, :-1] -= cov[:, :-1].mean(axis=0)
    cov[:, :-1] /= cov[:, :-1].std(axis=0)
    if reml:
        reml_d = REML(SparseCholesky(), [A], cov, y, verbose=True)
        print("reml d are %s and %s" % (reml_d["covariance coefficients"], reml_d["covariates coefficients"]))
        return reml_d
    he_est = HE([A], cov, y, compute_stderr=True, y2=y2)
    print("HE estimates are %s and %s" % (he_est[0], he_est[1]))
    return he_est


def read_relationship_matrix(path):
    """``mmread(path).tocsr()`` of the reference (SparseCholesky.py:399) -- natively parsed; ``.npz`` also accepted."""
    if str(path).endswith('.npz'):
        return sparse.load_npz(path).tocsr()
    with open(path, 'rb') as fh:
        banner = fh.readline(1024).decode('latin-1').lower()
    if banner.startswith('%%matrixmarket') and any(w in banner.split() for w in ('array', 'complex', 'hermitian')):
        # the two layouts the native parser does not take (dense array format, complex values) go to SciPy's general reader --
        # decided from the banner, never as a reaction to a parse error: a malformed coordinate file raises
        return sparse.csr_matrix(mmread(path))
    return _lib.read_matrix_market(path)


def run_estimates_from_paths(A, phe, cov, reml=False, ignore_indices=False):
    """File front end (SparseCholesky.py:398-403): MatrixMarket A, header-less phenotype CSV, covariate CSV.

    The MatrixMarket file goes through the native streaming parser (``csrc/mmio.cpp``: memory-mapped, all host
    cores) instead of ``scipy.io.mmread``; a ``.npz`` written by ``scipy.sparse.save_npz`` is accepted behind the
    same ``--A`` flag (binary, no parsing at all)."""
    A = read_relationship_matrix(A)
    index_col = None if ignore_indices else 0
    df_cov = pd.read_csv(cov, index_col=index_col)
    df_phe = pd.read_csv(phe, header=None, index_col=index_col)
    return run_estimates(A, df_phe, df_cov, reml=reml, ignore_indices=ignore_indices)


def _main(argv=None):
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--A', required=True, help="Path of the covariance matrix A (MatrixMarket).")
    parser.add_argument('--phe', required=True,
                        help="Phenotype CSV without header: IID,phenotype (only the phenotype with --ignore_indices).")
    parser.add_argument('--cov', required=True,
                        help="Covariates CSV with header; first column IID unless --ignore_indices.")
    parser.add_argument('--reml', default=False, action='store_true',
                        help="Compute using REML, default case uses the HE estimation method.")
    parser.add_argument('--ignore_indices', default=False, action='store_true',
                        help="Assume A, phenotypes and covariates are already in the same order.")
    args = parser.parse_args(argv)
    return run_estimates_from_paths(**(args.__dict__))


if __name__ == '__main__':
    _main()
