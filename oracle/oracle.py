"""ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Python face of the plain-C oracle (oracle/chol_oracle.c) plus three independent factor stand-ins that
all obey the reference's factor protocol (``factor(b)``, ``factor.L()``, ``factor.P()``,
``factor.logdet()`` -- reference scilmm/SparseCholesky.py:30,32,40,50,52,93,100):

* ``OracleFactor``  -- up-looking simplicial Cholesky in C, any permutation;
* ``DenseFactor``   -- LAPACK Cholesky of V[P][:,P] (n <~ 2e4), the stand-in used to drive the
                       reference's own Python when the golden vectors were made (oracle/make_golden.py);
* ``SuperLUFactor`` -- SciPy SuperLU in symmetric mode (third-party sparse direct solver, own ordering):
                       pins logdet and solves independently of any code in this repository.

The third-party arithmetic of the reference (scikit-sparse>=0.4.3 -> SuiteSparse CHOLMOD, unpinned) is
absent from this container; parity at that boundary is pinned through the uniqueness of the Cholesky
factor for a given permutation ("Oracle").
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.linalg as la
import scipy.sparse as sp
import scipy.sparse.linalg as sla

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("chol_oracle.c", "supernodal_cpu.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.oracle_factorize.restype = vp
        L.oracle_factorize.argtypes = [i32, vp, vp, vp, vp, C.POINTER(i32)]
        L.oracle_free.argtypes = [vp]
        L.oracle_nnz.restype = i64
        L.oracle_nnz.argtypes = [vp]
        L.oracle_export.argtypes = [vp, vp, vp, vp, vp]
        L.oracle_logdet.restype = C.c_double
        L.oracle_logdet.argtypes = [vp]
        L.oracle_solve.argtypes = [vp, vp, i32, vp]
        L.oracle_lmul.argtypes = [vp, vp, i32, vp]
        L.oracle_quadforms.argtypes = [i32, vp, vp, vp, C.c_int, vp, i32, vp]
        L.sncpu_set_blas.argtypes = [vp, vp, vp, vp]
        L.sncpu_factorize.restype = C.c_int
        L.sncpu_factorize.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
        L.sncpu_solve.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, vp]
        L.sncpu_lmul.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, vp, vp]
        L.sncpu_update_from.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
        L.sncpu_finish.restype = C.c_int
        L.sncpu_finish.argtypes = [i32, vp, vp, vp, vp]
        L.sncpu_fwd_front.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32]
        L.sncpu_bwd_front.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, vp]
        L.sncpu_lmul_front.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, vp, vp]
        L.sncpu_factorize_range.restype = C.c_int
        L.sncpu_factorize_range.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class NotPositiveDefinite(Exception):
    pass


class OracleFactor(object):
    """Factor protocol on top of chol_oracle.c; L L^T = V[P][:,P] with P given (default identity)."""

    def __init__(self, V, perm=None):
        V = sp.csr_matrix(V)
        V.sort_indices()
        self.n = n = V.shape[0]
        self._indptr = V.indptr.astype(np.int64)
        self._indices = V.indices.astype(np.int32)
        self._data = V.data.astype(np.float64)
        self._perm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        bad = C.c_int32(-1)
        self._h = lib().oracle_factorize(n, _p(self._indptr), _p(self._indices), _p(self._data),
                                         None if self._perm is None else _p(self._perm), C.byref(bad))
        if not self._h:
            raise NotPositiveDefinite(bad.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_free(self._h)
            self._h = None

    def __call__(self, b):
        b = np.asarray(b, dtype=np.float64)
        B = np.ascontiguousarray(b.reshape(self.n, -1))
        X = np.empty_like(B)
        lib().oracle_solve(self._h, _p(B), B.shape[1], _p(X))
        return X.reshape(b.shape)

    def lmul(self, R):
        R = np.asarray(R, dtype=np.float64)
        R2 = np.ascontiguousarray(R.reshape(self.n, -1))
        Z = np.empty_like(R2)
        lib().oracle_lmul(self._h, _p(R2), R2.shape[1], _p(Z))
        return Z.reshape(R.shape)

    def L(self):
        nnz = lib().oracle_nnz(self._h)
        colptr = np.empty(self.n + 1, np.int64)
        rowidx = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.float64)
        lib().oracle_export(self._h, _p(colptr), _p(rowidx), _p(val), None)
        return sp.csc_matrix((val, rowidx, colptr), shape=(self.n, self.n))

    def P(self):
        return np.arange(self.n) if self._perm is None else self._perm.astype(np.int64)

    def logdet(self):
        return lib().oracle_logdet(self._h)


class DenseFactor(object):
    """LAPACK Cholesky of V[P][:,P]; the unique factor for that P (n <~ 2e4)."""

    def __init__(self, V, perm=None):
        Vd = V.toarray() if sp.issparse(V) else np.asarray(V)
        self.n = Vd.shape[0]
        self._P = np.arange(self.n) if perm is None else np.asarray(perm)
        self._Ld = la.cholesky(Vd[np.ix_(self._P, self._P)], lower=True)

    def __call__(self, b):
        b = np.asarray(b, dtype=np.float64)
        x = la.cho_solve((self._Ld, True), b[self._P])
        out = np.empty_like(x)
        out[self._P] = x
        return out

    def L(self):
        return sp.csc_matrix(np.tril(self._Ld))

    def P(self):
        return self._P

    def logdet(self):
        return 2.0 * np.log(np.diag(self._Ld)).sum()


class SuperLUFactor(object):
    """SciPy SuperLU, symmetric mode, no pivoting: V = L U with U = D L^T  => chol = L sqrt(D)."""

    def __init__(self, V, permc_spec="MMD_AT_PLUS_A"):
        V = sp.csc_matrix(V)
        self.n = V.shape[0]
        self._lu = sla.splu(V, permc_spec=permc_spec, diag_pivot_thresh=0.0, options={"SymmetricMode": True})
        if not np.array_equal(self._lu.perm_r, self._lu.perm_c):
            raise NotPositiveDefinite("SuperLU pivoted: matrix not SPD enough for symmetric mode")
        self._d = self._lu.U.diagonal()
        if np.any(self._d <= 0):
            raise NotPositiveDefinite("non-positive pivot")

    def __call__(self, b):
        return self._lu.solve(np.asarray(b, dtype=np.float64))

    def L(self):
        return (self._lu.L @ sp.diags(np.sqrt(self._d))).tocsc()

    def P(self):
        return np.argsort(self._lu.perm_c)

    def logdet(self):
        return np.log(self._d).sum()


def cholesky(V, perm=None, kind="c"):
    """cholesky_func stand-in with the reference's calling convention (SparseCholesky.py:22-26)."""
    if kind == "c":
        return OracleFactor(V, perm)
    if kind == "dense":
        return DenseFactor(V, perm)
    if kind == "superlu":
        return SuperLUFactor(V)
    raise ValueError(kind)


def quadforms(A, U, lower_only=False):
    """out[c] = sum_i (A U)_ic U_ic  (SparseCholesky.py:65) via the C loop."""
    A = sp.csr_matrix(A)
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(A.shape[0], -1)
    out = np.empty(U.shape[1])
    ip, ix, dx = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)
    lib().oracle_quadforms(A.shape[0], _p(ip), _p(ix), _p(dx), int(lower_only), _p(U), U.shape[1], _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# BLAS-backed supernodal CPU baseline (oracle/supernodal_cpu.c)

def _capsule_ptr(capsule):
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
    C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    return C.pythonapi.PyCapsule_GetPointer(capsule, C.pythonapi.PyCapsule_GetName(capsule))


_blas_set = False


def _set_blas():
    global _blas_set
    if _blas_set:
        return
    from scipy.linalg import cython_blas, cython_lapack
    lib().sncpu_set_blas(_capsule_ptr(cython_blas.__pyx_capi__["dgemm"]),
                         _capsule_ptr(cython_blas.__pyx_capi__["dsyrk"]),
                         _capsule_ptr(cython_blas.__pyx_capi__["dtrsm"]),
                         _capsule_ptr(cython_lapack.__pyx_capi__["dpotrf"]))
    _blas_set = True


class SupernodalCPU(object):
    """CPU supernodal LL^T driven by a symbolic analysis (dict of arrays as returned by
    scilmm_amd.factor.Symbolic.arrays()); used as bench.py's cpu_baseline ("port")."""

    def __init__(self, sym_arrays, n):
        _set_blas()
        self.a = {k: np.ascontiguousarray(v) for k, v in sym_arrays.items()}
        self.n = n
        self.ns = len(self.a["sn_start"]) - 1
        self.Lx = np.zeros(int(self.a["sn_loff"][-1]))

    def assemble(self, pattern_values):
        """pattern_values: values of V in pattern-slot order (len nnz_pattern)."""
        self.Lx[:] = 0.0
        self.Lx[self.a["asm_dst"]] = pattern_values

    def factorize(self):
        a = self.a
        st = lib().sncpu_factorize(self.ns, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]),
                                   _p(a["upd_ptr"]), _p(a["upd_src"]), _p(a["upd_p0"]), _p(a["upd_p1"]), self.n,
                                   _p(self.Lx))
        if st != 0:
            raise NotPositiveDefinite(st - 1)

    def factorize_range(self, s0, s1):
        """Panels of supernodes [s0, s1) only; their descendants must be final in self.Lx (oracle/dist_cpu.py)."""
        a = self.a
        st = lib().sncpu_factorize_range(s0, s1, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]),
                                         _p(a["upd_ptr"]), _p(a["upd_src"]), _p(a["upd_p0"]), _p(a["upd_p1"]), self.n,
                                         _p(self.Lx))
        if st != 0:
            raise NotPositiveDefinite(st - 1)

    def logdet(self):
        return 2.0 * np.log(self.Lx[self.a["diag_dst"]]).sum()

    # ---- one front at a time (oracle/dist_cpu.py); self.a["sn_loff"] may be a RANK-LOCAL offset table
    def update_from(self, s, d_lo, d_hi):
        a = self.a
        lib().sncpu_update_from(s, d_lo, d_hi, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]),
                                _p(a["upd_ptr"]), _p(a["upd_src"]), _p(a["upd_p0"]), _p(a["upd_p1"]), self.n, _p(self.Lx))

    def finish(self, s):
        a = self.a
        st = lib().sncpu_finish(s, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_loff"]), _p(self.Lx))
        if st != 0:
            raise NotPositiveDefinite(st - 1)

    def fwd_front(self, s, Y, ACC, push):
        a = self.a
        lib().sncpu_fwd_front(s, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]), _p(self.Lx), self.n,
                              Y.shape[1], _p(Y), _p(ACC), int(push))

    def bwd_front(self, s, Y):
        a = self.a
        lib().sncpu_bwd_front(s, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]), _p(self.Lx), self.n,
                              Y.shape[1], _p(Y))

    def lmul_front(self, s, R, Y):
        a = self.a
        lib().sncpu_lmul_front(s, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]), _p(self.Lx), self.n,
                               R.shape[1], _p(R), _p(Y))

    def L_csc(self):
        """The factor as scipy CSC (permuted labels), assembled from the panels."""
        a = self.a
        rows, cols, vals = [], [], []
        for s in range(self.ns):
            c0, w = int(a["sn_start"][s]), int(a["sn_start"][s + 1] - a["sn_start"][s])
            rs = a["sn_rows"][a["sn_rowptr"][s]:a["sn_rowptr"][s + 1]]
            m = rs.size
            P = self.Lx[a["sn_loff"][s]:a["sn_loff"][s] + m * w].reshape(w, m)  # column-major m x w
            for j in range(w):
                rows.append(rs[j:])
                cols.append(np.full(m - j, c0 + j))
                vals.append(P[j, j:])
        return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n, self.n))

    def solve_permuted(self, Y):
        """Y: (n, r) Fortran-ordered, already permuted; solved in place."""
        assert Y.flags.f_contiguous
        a = self.a
        lib().sncpu_solve(self.ns, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]),
                          _p(self.Lx), self.n, Y.shape[1], _p(Y))
        return Y

    def lmul(self, R):
        """(L R)[argsort(P)] (simulate_vector, SparseCholesky.py:50-51): R is not permuted on the way in."""
        a = self.a
        R = np.asarray(R, dtype=np.float64)
        Rf = np.asfortranarray(R.reshape(self.n, -1))
        Y = np.empty_like(Rf)
        lib().sncpu_lmul(self.ns, _p(a["sn_start"]), _p(a["sn_rowptr"]), _p(a["sn_rows"]), _p(a["sn_loff"]), _p(self.Lx),
                         self.n, Rf.shape[1], _p(Rf), _p(Y))
        Z = np.empty((self.n, Rf.shape[1]))
        Z[a["perm"]] = Y
        return Z.reshape(R.shape)

    def solve(self, B):
        perm = self.a["perm"]
        B = np.asarray(B, dtype=np.float64)
        Y = np.asfortranarray(B.reshape(self.n, -1)[perm])
        self.solve_permuted(Y)
        X = np.empty((self.n, Y.shape[1]))
        X[perm] = Y
        return X.reshape(B.shape)


class CPUPortFactor(object):
    """Factor protocol (``factor(b)``, ``.L()``, ``.P()``, ``.logdet()`` + ``lmul``) on top of ``SupernodalCPU`` for a
    given analysis: the CPU counterpart of the HIP factor at the sizes the simplicial C oracle cannot reach (100k
    config: 1.7 TFLOP).  ``sym_arrays`` = ``Symbolic.arrays()`` (+ ``pat_colptr``) of the engine under test, so both
    factor V[P][:,P] for the SAME P with the same supernode blocks -- through entirely different code (BLAS-3 on the
    host vs. HIP kernels)."""

    def __init__(self, sym_arrays, pat_colptr, V):
        V = sp.csr_matrix(V)
        self.n = n = V.shape[0]
        self.cpu = SupernodalCPU(sym_arrays, n)
        self._perm = np.asarray(sym_arrays["perm"])
        p = self._perm
        Lw = sp.tril(V[p][:, p]).tocsc()
        Lw.sort_indices()
        if not np.array_equal(Lw.indptr, pat_colptr):
            raise ValueError("V does not have the analysed pattern")
        self.cpu.assemble(Lw.data)  # pattern slots = CSC order of tril(V[P][:,P])
        self.cpu.factorize()
        self._L = None

    def __call__(self, b):
        return self.cpu.solve(b)

    def L(self):
        if self._L is None:
            self._L = self.cpu.L_csc()
        return self._L

    def lmul(self, R):
        return self.cpu.lmul(R)

    def P(self):
        return self._perm.astype(np.int64)

    def logdet(self):
        return self.cpu.logdet()


def dominance(parents, A):
    """CPU restatement of reference scilmm/Matrices/Dominance.py:12-43 (test infrastructure, like everything here).

    D_ij = 0.25 * (A[f_i,f_j] * A[m_i,m_j] + A[f_i,m_j] * A[m_i,f_j]) on the stored pattern of A (explicit zeros removed
    first, :14), unit diagonal (:41-42); an unknown parent (-1) indexes the reference's all-zero root row (:17-24) and
    contributes 0.  Same operation order as the reference's NumPy expression (:28-32), so the values are its bits.
    """
    import scipy.sparse as sp
    A = sp.csr_matrix(A).copy()
    A.eliminate_zeros()
    A.sort_indices()
    n = A.shape[0]
    par = np.asarray(parents).reshape(n, 2).astype(np.int64)
    ii = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr))
    jj = A.indices.astype(np.int64)
    keys = ii * n + jj  # ascending: rows ascending, columns ascending inside a row

    def look(r, c):
        out = np.zeros(r.size)
        ok = (r >= 0) & (c >= 0)
        k = r[ok] * n + c[ok]
        pos = np.minimum(np.searchsorted(keys, k), keys.size - 1)
        out[ok] = np.where(keys[pos] == k, A.data[pos], 0.0)
        return out

    fi, mi, fj, mj = par[ii, 0], par[ii, 1], par[jj, 0], par[jj, 1]
    vals = look(fi, fj) * look(mi, mj) + look(fi, mj) * look(mi, fj)
    vals *= 0.25
    vals[ii == jj] = 1.0
    D = sp.csr_matrix((vals, A.indices.copy(), A.indptr.copy()), shape=A.shape)
    D.eliminate_zeros()  # the reference's closing sparse arithmetic (:41-43) drops the entries that came out as 0
    return D
