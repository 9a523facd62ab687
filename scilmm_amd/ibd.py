"""Native pedigree -> IBD builder (the step before the hot path).

``ibd_from_parents`` / ``count_ibd_nonzero`` have the semantics of the reference's ``Numerator.LD`` +
``create_numerator`` (scilmm/Matrices/Numerator.py:5-38) and ``Relationship.count_IBD_nonzero``
(scilmm/Matrices/Relationship.py:38-61) but run as C++/OpenMP behind the C-ABI (csrc/ibd.cpp).
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from ._lib import check, lib, ptr


def _parents32(par):
    par = np.ascontiguousarray(par, dtype=np.int32)
    if par.ndim != 2 or par.shape[1] != 2:
        raise ValueError("parents must be an (n, 2) table, -1 = unknown")
    return par


def count_ibd_nonzero(par):
    """Number of structural nonzeros of A (both triangles) without building it."""
    par = _parents32(par)
    h, nnz = C.c_void_p(), C.c_int64(0)
    check(lib().scilmm_ibd_build(par.shape[0], ptr(par), 1, C.byref(h), C.byref(nnz)))
    lib().scilmm_ibd_free(h)
    return nnz.value


def ibd_pattern_from_parents(par, values=True):
    """The PATTERN of A (pairs with a common ancestor; both triangles, sorted) as a CSR matrix of ones -- no values are
    computed: they can be produced on the device (``Symbolic.ibd_values_from_pedigree``).  ``values=False``: a
    ``scilmm_amd.factor.PatternCSR`` (no value array at all: 8 bytes per entry less on the host)."""
    par = _parents32(par)
    n = par.shape[0]
    h, nnz = C.c_void_p(), C.c_int64(0)
    check(lib().scilmm_ibd_build(n, ptr(par), 2, C.byref(h), C.byref(nnz)))
    try:
        ap, ai = np.empty(n + 1, np.int64), np.empty(nnz.value, np.int32)
        check(lib().scilmm_ibd_export(h, ptr(ap), ptr(ai), None, None, None, None, None, None))
    finally:
        lib().scilmm_ibd_free(h)
    if not values:
        from .factor import PatternCSR
        return PatternCSR(ap, ai, n)
    return sp.csr_matrix((np.ones(nnz.value), ai, ap), shape=(n, n))


def ibd_from_parents(par, return_LD=False):
    """A = L D L^T (symmetric CSR, sorted); optionally also L (CSR) and D (vector)."""
    par = _parents32(par)
    n = par.shape[0]
    h, nnz = C.c_void_p(), C.c_int64(0)
    check(lib().scilmm_ibd_build(n, ptr(par), 0, C.byref(h), C.byref(nnz)))
    try:
        na, nl = C.c_int64(0), C.c_int64(0)
        check(lib().scilmm_ibd_sizes(h, C.byref(na), C.byref(nl)))
        ap, ai, ax = np.empty(n + 1, np.int64), np.empty(na.value, np.int32), np.empty(na.value)
        if return_LD:
            lp, li, lx, D = np.empty(n + 1, np.int64), np.empty(nl.value, np.int32), np.empty(nl.value), np.empty(n)
            check(lib().scilmm_ibd_export(h, ptr(ap), ptr(ai), ptr(ax), ptr(lp), ptr(li), ptr(lx), ptr(D), None))
        else:
            check(lib().scilmm_ibd_export(h, ptr(ap), ptr(ai), ptr(ax), None, None, None, None, None))
    finally:
        lib().scilmm_ibd_free(h)
    A = sp.csr_matrix((ax, ai, ap), shape=(n, n))
    if return_LD:
        return A, sp.csr_matrix((lx, li, lp), shape=(n, n)), D
    return A
