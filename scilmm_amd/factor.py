"""Python handles over the C-ABI (include/scilmm_hip.h): ``Symbolic`` (pattern analysis + device-resident
A_k values) and ``Factor`` (numeric factor with the reference's factor protocol).

Factor protocol mirrored from what the reference touches on an sksparse Factor
(reference scilmm/SparseCholesky.py:30,32,40,50,52,93,100; scilmm/Estimation/LMM.py:28,30,38,48-50,87,94):
``factor(b)``, ``factor.L()``, ``factor.P()``, ``factor.logdet()`` -- plus ``factor.lmul(R)``, the fused
form of ``factor.L().dot(R)[argsort(P)]`` that never materialises L on the host.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import check, lib, ptr

_ORDERINGS = {"amd": 0, "natural": 1, "given": 2, "nesdis": 3, "best": 4}


class PatternCSR(object):
    """A CSR sparsity pattern WITHOUT values (canonical: rows ascending, both halves stored): what ``Symbolic`` needs of a
    matrix whose values are then computed on the device (``ibd_values_from_pedigree``, ``dominance_values_from``) -- at
    BASELINE configs[4]'s 3M pedigree that is 21.6 GB of values per matrix that never exist on the host."""

    has_sorted_indices = True
    data = None

    def __init__(self, indptr, indices, n):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.shape = (int(n), int(n))
        if self.indptr.shape != (n + 1,) or int(self.indptr[-1]) != self.indices.size:
            raise ValueError("indptr / indices do not describe an n x n CSR pattern")
        self.nnz = int(self.indices.size)


def _as_csr(m):
    if isinstance(m, PatternCSR):
        return m
    m = sp.csr_matrix(m)
    if not m.has_sorted_indices:
        m = m.sorted_indices()
    return m


class Symbolic(object):
    """Symbolic analysis of the pattern of V = sum_k s2_k mats[k]; built once, reused for every sigma2."""

    def __init__(self, mats, perm=None, ordering="amd", upload=True, **opts):
        mats = [_as_csr(m) for m in mats]
        self.n = n = mats[0].shape[0]
        for m in mats:
            if m.shape != (n, n):
                raise ValueError("all matrices must be square and of equal shape")
        self.K = K = len(mats)
        self._indptr = [np.ascontiguousarray(m.indptr, dtype=np.int64) for m in mats]
        self._indices = [np.ascontiguousarray(m.indices, dtype=np.int32) for m in mats]
        self._data = [None if m.data is None else np.ascontiguousarray(m.data, dtype=np.float64) for m in mats]
        self._values_epoch = 0  # bumped whenever the resident values of a matrix change (Factor.holds)
        if perm is not None:
            ordering = "given"
            perm = np.ascontiguousarray(perm, dtype=np.int32)
            if perm.shape != (n,):
                raise ValueError("perm must have length n")
        cache = opts.pop("cache", None) or os.environ.get("SCILMM_SYMBOLIC_CACHE")
        o = _lib.Options.default(_ORDERINGS[ordering], **opts)
        a = (C.c_void_p * K)(*[x.ctypes.data for x in self._indptr])
        b = (C.c_void_p * K)(*[x.ctypes.data for x in self._indices])
        h = C.c_void_p()
        self.from_cache = False
        path = key = None
        if cache:
            # image of the analysis keyed by everything it depends on: patterns, permutation, ordering, options
            # (scilmm_symbolic_save / _load: once per pattern and node instead of once per process)
            key = self._analysis_key(n, perm, ordering, opts)
            path = os.path.join(cache, "scilmm_symbolic_%016x.bin" % key)
            if lib().scilmm_symbolic_load(os.fsencode(path), C.c_uint64(key), C.byref(h)) == _lib.OK:
                self.from_cache = True
        if not self.from_cache:
            st = lib().scilmm_symbolic_create(n, K, a, b, None if perm is None else ptr(perm), C.byref(o), 1, C.byref(h))
            self._h = h
            check(st, h)
            if path is not None:
                os.makedirs(cache, exist_ok=True)
                lib().scilmm_symbolic_save(h, os.fsencode(path), C.c_uint64(key))  # best effort: a failed write is not an error
        self._h = h
        self._uploaded = False
        self.front_bits = 64
        if upload:
            self.upload_values()

    # every environment variable the host analysis reads (csrc/symbolic.cpp, csrc/capi_symbolic.cpp): part of the cache key
    _ANALYSIS_ENV = ("SCILMM_TUNING", "SCILMM_TAIL_WIDE", "SCILMM_TAIL_ELIG", "SCILMM_TAIL_DELAY")
    # bumped whenever the analysis changes what it produces for the same input (the library's magic number guards the FORMAT of
    # the image, this constant and scilmm_version() its CONTENT)
    _ANALYSIS_REVISION = 4

    def _analysis_key(self, n, perm, ordering, opts):
        """64-bit key of everything the analysis depends on: patterns, permutation, ordering, options, the environment
        variables it reads, the library version and the analysis revision.  xxhash when it is installed (0.5 s per 4 GB of
        pattern), hashlib.blake2b otherwise (standard library: no dependency is needed for the cache to work)."""
        try:
            import xxhash
            hx = xxhash.xxh64()
            digest = hx.intdigest
        except ImportError:
            import hashlib
            hx = hashlib.blake2b(digest_size=8)
            digest = lambda: int.from_bytes(hx.digest(), "little")
        hx.update(np.array([n, self.K, _ORDERINGS[ordering], self._ANALYSIS_REVISION], dtype=np.int64).tobytes())
        hx.update(lib().scilmm_version())
        hx.update(repr(sorted(opts.items())).encode())
        hx.update(repr(sorted((k, v) for k, v in os.environ.items() if k in self._ANALYSIS_ENV)).encode())
        if perm is not None:
            hx.update(perm.tobytes())
        for ip, ix in zip(self._indptr, self._indices):
            hx.update(memoryview(ip).cast("B"))
            hx.update(memoryview(ix).cast("B"))
        return digest()

    def upload_values(self, skip=()):
        self._values_epoch += 1
        for k in range(self.K):
            if k not in skip and self._data[k] is not None:  # (None: a PatternCSR -- its values are built on the device)
                check(lib().scilmm_values_upload(self._h, k, ptr(self._data[k])), self._h)
        self._uploaded = True

    def ibd_values_from_pedigree(self, k, parents):
        """Compute matrix k's values -- the IBD (numerator relationship) matrix of the pedigree ``parents`` ((n, 2), -1 =
        unknown, individuals in the matrices' row order, parents before children) -- ON THE DEVICE, straight into its
        HBM-resident value slots: they never cross PCIe (``scilmm_ibd_values_device``).  The analysed pattern of matrix k
        must be the pedigree's common-ancestor pattern (``scilmm_amd.ibd.ibd_pattern_from_parents``)."""
        par = np.ascontiguousarray(parents, dtype=np.int32)
        if par.shape != (self.n, 2):
            raise ValueError("parents must be an (n, 2) table")
        self._values_epoch += 1
        check(lib().scilmm_ibd_values_device(self._h, k, self.n, ptr(par)), self._h)

    def dominance_values_from(self, k_dst, k_src, parents):
        """Compute matrix ``k_dst``'s values -- the dominance relationship matrix of the pedigree ``parents`` on the analysed
        pattern (reference scilmm/Matrices/Dominance.py:12-43) -- ON THE DEVICE from matrix ``k_src``'s resident IBD values,
        straight into its value slots (``scilmm_dominance_values_device``)."""
        par = np.ascontiguousarray(parents, dtype=np.int32)
        if par.shape != (self.n, 2):
            raise ValueError("parents must be an (n, 2) table")
        self._values_epoch += 1
        check(lib().scilmm_dominance_values_device(self._h, k_dst, k_src, self.n, ptr(par)), self._h)

    def values_slots(self, k):
        """Matrix k's device-resident values in pattern-slot order (tests, diagnostics)."""
        diag_only = self._indices[k].size == self.n and np.array_equal(self._indices[k], np.arange(self.n))
        out = np.empty(self.n if diag_only else self.info().nnz_pattern)
        check(lib().scilmm_values_download(self._h, k, ptr(out)), self._h)
        return out

    def release_host_maps(self):
        """Free the host copies of the value-assembly maps and of the uploaded values (a process that only evaluates does not
        need them: 13 + 7 GB at the 1M config); ``set_values`` is no longer available afterwards."""
        check(lib().scilmm_symbolic_release_host_maps(self._h), self._h)
        self._data = [None] * self.K

    def set_values(self, k, data):
        """Replace the values of matrix k (same pattern)."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        if self._data[k] is not None:
            if data.shape != self._data[k].shape:
                raise ValueError("value array does not match the analysed pattern")
            self._data[k] = data
        self._values_epoch += 1
        check(lib().scilmm_values_upload(self._h, k, ptr(data)), self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().scilmm_symbolic_free(h)
            except Exception:  # interpreter shutdown: the module's globals are already gone, the process frees everything
                pass
            self._h = None

    def info(self):
        info = _lib.Info()
        check(lib().scilmm_symbolic_info(self._h, C.byref(info)), self._h)
        return info

    def get(self, name):
        return _lib.symbolic_get(self._h, name)

    def arrays(self):
        names = ["perm", "sn_start", "sn_rowptr", "sn_rows", "sn_loff", "upd_ptr", "upd_src", "upd_p0", "upd_p1",
                 "asm_dst", "diag_dst"]
        return {k: self.get(k) for k in names}

    def P(self):
        return self.get("perm").astype(np.int64)

    def factorize(self, sigma2):
        return Factor(self, sigma2)

    def quadforms(self, k, U):
        """out[c] = sum_i (A_k U)_ic U_ic  (compute_gradients, SparseCholesky.py:65)."""
        U = np.ascontiguousarray(np.asarray(U, dtype=np.float64).reshape(self.n, -1))
        out = np.empty(U.shape[1])
        check(lib().scilmm_quadforms(self._h, k, ptr(U), U.shape[1], ptr(out)), self._h)
        return out

    def he_moments(self, k1, k2):
        """(sum(A_k1 o A_k2), diag(A_k1) . diag(A_k2)) from the device-resident values (HE, SparseCholesky.py:223-231)."""
        fro, dg = C.c_double(0.0), C.c_double(0.0)
        check(lib().scilmm_he_moments(self._h, k1, k2, C.byref(fro), C.byref(dg)), self._h)
        return fro.value, dg.value

    def spmm(self, k, X):
        X = np.asarray(X, dtype=np.float64)
        X2 = np.ascontiguousarray(X.reshape(self.n, -1))
        Y = np.empty_like(X2)
        check(lib().scilmm_spmm(self._h, k, ptr(X2), X2.shape[1], ptr(Y)), self._h)
        return Y.reshape(X.shape)

    def set_front_precision(self, bits):
        """32: dense-tail products on the fp32 matrix pipe, sums in fp64 (BASELINE configs[4]); 64: all fp64 (default).
        With 32 the factor has a ~1e-7 relative backward error; ``Factor.__call__`` then refines its solves."""
        check(lib().scilmm_set_front_precision(self._h, int(bits)), self._h)
        self.front_bits = int(bits)

    def set_profiling(self, on=True):
        """True / 1: HIP-event brackets per kernel class; 2: also queue the look-ahead launches on one stream (clean
        per-launch durations of the dominant kernel; a measurement mode)."""
        check(lib().scilmm_set_profiling(self._h, 2 if on == 2 else int(bool(on))), self._h)

    def sync(self):
        check(lib().scilmm_sync(self._h), self._h)

    def spmm_dev(self, k, dX_ptr, r, dY_ptr):
        """Y = A_k X with device pointers ([n][r] blocks), asynchronous on the engine's stream."""
        check(lib().scilmm_spmm_dev(self._h, k, dX_ptr, r, dY_ptr), self._h)

    def quadforms_dev(self, k, dU_ptr, r, dout_ptr):
        check(lib().scilmm_quadforms_dev(self._h, k, dU_ptr, r, dout_ptr), self._h)

    def timing(self):
        t = _lib.Timing()
        check(lib().scilmm_last_timing(self._h, C.byref(t)), self._h)
        return {f: getattr(t, f) for f, _ in _lib.Timing._fields_}


class Factor(object):
    """Numeric factor L L^T = V[P][:,P] living in HBM."""

    def __init__(self, symbolic, sigma2):
        self.sym = symbolic
        self.n = symbolic.n
        s2 = np.ascontiguousarray(sigma2, dtype=np.float64)
        if s2.shape != (symbolic.K,):
            raise ValueError("sigma2 must have one entry per matrix")
        h = C.c_void_p()
        bad = C.c_int32(-1)
        st = lib().scilmm_factorize(symbolic._h, ptr(s2), C.byref(h), C.byref(bad))
        self._h = h
        self._s2, self._epoch, self._pending = None, -1, None
        check(st, symbolic._h, bad.value)
        self._s2, self._epoch = s2.copy(), symbolic._values_epoch

    @classmethod
    def from_handle(cls, symbolic, handle):
        """A Factor object around an existing ``scilmm_factor*`` (caller-owned storage: ``scilmm_factor_create_external``,
        the multi-GPU engine); nothing is factorized yet."""
        f = cls.__new__(cls)
        f.sym, f.n, f._h = symbolic, symbolic.n, handle
        f._s2, f._epoch, f._pending = None, -1, None
        return f

    def holds(self, sigma2):
        """True when this object IS the factor of sum_k sigma2[k] A_k for the values now resident (so that a caller -- the
        reference factorizes once more at the optimum, SparseCholesky.py:182 -- need not repeat a factorization it has)."""
        return (self._s2 is not None and self._epoch == self.sym._values_epoch
                and np.array_equal(self._s2, np.asarray(sigma2, dtype=np.float64)))

    def refactorize(self, sigma2):
        s2 = np.ascontiguousarray(sigma2, dtype=np.float64)
        bad = C.c_int32(-1)
        self._s2, self._pending = None, None  # (a recorded async request is superseded: its sigma2 must never be adopted later)
        check(lib().scilmm_refactorize(self._h, ptr(s2), C.byref(bad)), self.sym._h, bad.value)
        self._s2, self._epoch = s2.copy(), self.sym._values_epoch
        return self

    def refactorize_async(self, sigma2):
        """Queue the refactorization and return; ``wait()`` (or any use of the factor) completes it."""
        s2 = np.ascontiguousarray(sigma2, dtype=np.float64)
        self._s2, self._pending = None, None
        check(lib().scilmm_refactorize_async(self._h, ptr(s2)), self.sym._h)
        self._pending = (s2.copy(), self.sym._values_epoch)
        return self

    def wait(self):
        bad = C.c_int32(-1)
        pending, self._pending = getattr(self, "_pending", None), None  # cleared whatever happens: a failed factorization
        check(lib().scilmm_factor_wait(self._h, C.byref(bad)), self.sym._h, bad.value)  # (raises) leaves no sigma2 behind
        if pending is not None:
            self._s2, self._epoch = pending
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().scilmm_factor_free(h)
            except Exception:  # (interpreter shutdown, as above)
                pass
            self._h = None

    def _rhs(self, fn, b):
        b = np.asarray(b, dtype=np.float64)
        if b.shape[0] != self.n:
            raise ValueError("right-hand side has %d rows, expected %d" % (b.shape[0], self.n))
        B = np.ascontiguousarray(b.reshape(self.n, -1))
        X = np.empty_like(B)
        check(fn(self._h, ptr(B), B.shape[1], ptr(X)), self.sym._h)
        return X.reshape(b.shape)

    REFINE_STEPS = 2  # iterative-refinement sweeps of a solve on a factor with fp32 fronts

    def __call__(self, b):
        """factor(b) = V^{-1} b for b of shape (n,) or (n, r).  On a factor with fp32 fronts (set_front_precision(32))
        the solve is refined against the exact V = sum_k s2_k A_k (fp64 SpMM on the device): each sweep gains ~7 digits."""
        if getattr(self, "_pending", None) is not None:
            self.wait()
        x = self._rhs(lib().scilmm_solve, b)
        if getattr(self.sym, "front_bits", 64) == 32 and getattr(self, "_s2", None) is not None:
            b = np.asarray(b, dtype=np.float64)
            for _ in range(self.REFINE_STEPS):
                r = b.copy()
                for k in range(self.sym.K):
                    r -= self._s2[k] * self.sym.spmm(k, x)
                x = x + self._rhs(lib().scilmm_solve, r)
        return x

    def lmul(self, R):
        """(factor.L() @ R)[argsort(factor.P())] without exporting L (simulate_vector, SparseCholesky.py:50-51)."""
        return self._rhs(lib().scilmm_lmul, R)

    def solve_dev(self, dB_ptr, r, dX_ptr):
        """Device-pointer solve (row-major n x r, original row order); enqueues without synchronising."""
        check(lib().scilmm_solve_dev(self._h, dB_ptr, r, dX_ptr), self.sym._h)

    def lmul_dev(self, dR_ptr, r, dZ_ptr):
        check(lib().scilmm_lmul_dev(self._h, dR_ptr, r, dZ_ptr), self.sym._h)

    def inverse_traces(self):
        """tr(V^-1 A_k) for every matrix of the analysis, exactly: the selected inverse on the supernodal factor (Takahashi
        recursion on the device, in place) followed by one pass over each A_k's pattern.  CONSUMES the factor: it must be
        refactorized before the next solve."""
        if getattr(self, "_pending", None) is not None:
            self.wait()
        self._s2, self._pending = None, None  # (the panels hold entries of the inverse from here on)
        check(lib().scilmm_selected_inverse(self._h), self.sym._h)
        out = np.empty(self.sym.K)
        check(lib().scilmm_inverse_traces(self._h, ptr(out)), self.sym._h)
        return out

    def logdet(self):
        out = C.c_double(0.0)
        check(lib().scilmm_logdet(self._h, C.byref(out)), self.sym._h)
        return out.value

    def P(self):
        return self.sym.P()

    def L(self):
        nnz = C.c_int64(0)
        check(lib().scilmm_export_L(self._h, None, None, None, C.byref(nnz)), self.sym._h)
        colptr = np.empty(self.n + 1, np.int64)
        rowidx = np.empty(nnz.value, np.int32)
        vals = np.empty(nnz.value, np.float64)
        check(lib().scilmm_export_L(self._h, ptr(colptr), ptr(rowidx), ptr(vals), C.byref(nnz)), self.sym._h)
        return sp.csc_matrix((vals, rowidx, colptr), shape=(self.n, self.n))
