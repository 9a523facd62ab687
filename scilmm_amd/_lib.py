"""ctypes binding of libscilmm_hip.so (include/scilmm_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a device call fails, an
exception is raised.  (The CPU oracle lives under oracle/ and is only used by tests and bench.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCILMM_HIP_LIB", os.path.join(_HERE, "csrc", "libscilmm_hip.so"))

OK, ERR_ARG, ERR_NOT_PD, ERR_DEVICE, ERR_STATE = 0, -1, -2, -3, -4


class ScilmmError(RuntimeError):
    pass


class NotPositiveDefiniteError(ScilmmError):
    """Counterpart of sksparse.cholmod.CholmodNotPositiveDefiniteError at the reference boundary."""

    def __init__(self, column):
        super().__init__("matrix is not positive definite (failing permuted column %d)" % column)
        self.column = column


class Options(C.Structure):
    _fields_ = [("ordering", C.c_int32), ("relax_small", C.c_int32), ("relax_w1", C.c_int32),
                ("relax_w2", C.c_int32), ("relax_z1", C.c_double), ("relax_z2", C.c_double),
                ("relax_z3", C.c_double), ("amd_dense", C.c_double), ("max_width", C.c_int32),
                ("nd_oksep", C.c_double), ("dense_relax", C.c_double)]

    @classmethod
    def default(cls, ordering=0, **kw):
        o = cls(ordering, -1, -1, -1, -1.0, -1.0, -1.0, 0.0, 0, 0.0, 0.0)
        for k, v in kw.items():
            setattr(o, k, v)
        return o


class Info(C.Structure):
    _fields_ = [("n", C.c_int32), ("K", C.c_int32), ("nsuper", C.c_int32), ("nlevels", C.c_int32),
                ("nnzL", C.c_int64), ("nnzL_stored", C.c_int64), ("nnz_pattern", C.c_int64),
                ("flops", C.c_double), ("n_rows_total", C.c_int64), ("n_updates", C.c_int64),
                ("update_flops", C.c_double), ("solve_flops_per_rhs", C.c_double),
                ("update_flops_executed", C.c_double), ("dense_first", C.c_int32), ("dense_flops", C.c_double)]


class Timing(C.Structure):
    _fields_ = [("assemble_ms", C.c_double), ("factor_ms", C.c_double), ("solve_fwd_ms", C.c_double),
                ("solve_bwd_ms", C.c_double), ("lmul_ms", C.c_double), ("quad_ms", C.c_double),
                ("n_launches", C.c_int64), ("update_ms", C.c_double), ("potrf_ms", C.c_double),
                ("trsm_ms", C.c_double), ("n_update_launches", C.c_int64), ("reduce_cells_ms", C.c_double),
                ("update_union_ms", C.c_double), ("dense_ms", C.c_double), ("n_dense_launches", C.c_int64),
                ("n_late_split", C.c_int64)]


# every symbol include/scilmm_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "scilmm_symbolic_create", "scilmm_symbolic_info", "scilmm_symbolic_get", "scilmm_symbolic_error",
    "scilmm_symbolic_free", "scilmm_symbolic_save", "scilmm_symbolic_load", "scilmm_symbolic_release_host_maps", "scilmm_values_upload", "scilmm_factorize", "scilmm_refactorize",
    "scilmm_refactorize_async", "scilmm_factor_wait", "scilmm_factor_free", "scilmm_logdet", "scilmm_solve", "scilmm_lmul", "scilmm_export_L",
    "scilmm_quadforms", "scilmm_spmm", "scilmm_spmm_dev", "scilmm_solve_dev", "scilmm_lmul_dev", "scilmm_quadforms_dev",
    "scilmm_sync", "scilmm_last_timing", "scilmm_set_profiling", "scilmm_version",
    "scilmm_ibd_build", "scilmm_ibd_sizes", "scilmm_ibd_export", "scilmm_ibd_free", "scilmm_ibd_values_device",
    "scilmm_dominance_values_device", "scilmm_values_download",
    "scilmm_order", "scilmm_fill_count",
    "scilmm_dist_init", "scilmm_dist_work_size", "scilmm_dist_set_work", "scilmm_dist_layout", "scilmm_factor_sizes", "scilmm_factor_create_external", "scilmm_he_moments", "scilmm_set_front_precision",
    "scilmm_selected_inverse", "scilmm_inverse_traces",
    "scilmm_mm_read", "scilmm_mm_export", "scilmm_mm_error", "scilmm_mm_free",
    "scilmm_dominance", "scilmm_dominance_dev", "scilmm_dominance_error",
    "scilmm_csr_spmm_dev", "scilmm_csr_spmm_error",
]

_lib = None


def lib():
    """Load the shared library (once). Raises ScilmmError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScilmmError("libscilmm_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (or make -C scilmm_amd/csrc); there is no CPU fallback" % LIB_PATH)
    # torch (device buffers of the REML evaluation, torch.distributed) ships its own copy of the HIP runtime: if this
    # library pulled /opt/rocm's in first, a later `import torch` would find no GPU.  Loading torch first makes both
    # sides share one runtime (the order bench.py always had).  SCILMM_NO_TORCH=1 skips it (host buffers only).
    if os.environ.get("SCILMM_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    L.scilmm_symbolic_create.argtypes = [i32, i32, P(vp), P(vp), vp, P(Options), i32, P(vp)]
    L.scilmm_symbolic_info.argtypes = [vp, P(Info)]
    L.scilmm_symbolic_get.argtypes = [vp, C.c_char_p, vp, P(i64)]
    L.scilmm_symbolic_error.argtypes = [vp]
    L.scilmm_symbolic_error.restype = C.c_char_p
    L.scilmm_symbolic_free.argtypes = [vp]
    L.scilmm_symbolic_free.restype = None
    L.scilmm_symbolic_save.argtypes = [vp, C.c_char_p, C.c_uint64]
    L.scilmm_symbolic_load.argtypes = [C.c_char_p, C.c_uint64, P(vp)]
    L.scilmm_symbolic_release_host_maps.argtypes = [vp]
    L.scilmm_values_upload.argtypes = [vp, i32, vp]
    L.scilmm_factorize.argtypes = [vp, vp, P(vp), P(i32)]
    L.scilmm_refactorize.argtypes = [vp, vp, P(i32)]
    L.scilmm_refactorize_async.argtypes = [vp, vp]
    L.scilmm_factor_wait.argtypes = [vp, P(i32)]
    L.scilmm_factor_free.argtypes = [vp]
    L.scilmm_factor_free.restype = None
    L.scilmm_logdet.argtypes = [vp, P(dbl)]
    L.scilmm_solve.argtypes = [vp, vp, i32, vp]
    L.scilmm_lmul.argtypes = [vp, vp, i32, vp]
    L.scilmm_export_L.argtypes = [vp, vp, vp, vp, P(i64)]
    L.scilmm_quadforms.argtypes = [vp, i32, vp, i32, vp]
    L.scilmm_spmm.argtypes = [vp, i32, vp, i32, vp]
    L.scilmm_spmm_dev.argtypes = [vp, i32, vp, i32, vp]
    L.scilmm_solve_dev.argtypes = [vp, vp, i32, vp]
    L.scilmm_lmul_dev.argtypes = [vp, vp, i32, vp]
    L.scilmm_quadforms_dev.argtypes = [vp, i32, vp, i32, vp]
    L.scilmm_sync.argtypes = [vp]
    L.scilmm_last_timing.argtypes = [vp, P(Timing)]
    L.scilmm_set_profiling.argtypes = [vp, i32]
    L.scilmm_version.restype = C.c_char_p
    L.scilmm_ibd_build.argtypes = [i32, vp, i32, P(vp), P(i64)]
    L.scilmm_ibd_sizes.argtypes = [vp, P(i64), P(i64)]
    L.scilmm_ibd_export.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.scilmm_ibd_free.argtypes = [vp]
    L.scilmm_ibd_values_device.argtypes = [vp, i32, i32, vp]
    L.scilmm_values_download.argtypes = [vp, i32, vp]
    L.scilmm_dominance_values_device.argtypes = [vp, i32, i32, i32, vp]
    L.scilmm_ibd_free.restype = None
    L.scilmm_mm_read.argtypes = [C.c_char_p, P(vp), P(i32), P(i32), P(i64)]
    L.scilmm_mm_export.argtypes = [vp, vp, vp, vp]
    L.scilmm_mm_error.argtypes = [vp]
    L.scilmm_mm_error.restype = C.c_char_p
    L.scilmm_mm_free.argtypes = [vp]
    L.scilmm_mm_free.restype = None
    L.scilmm_set_front_precision.argtypes = [vp, i32]
    L.scilmm_selected_inverse.argtypes = [vp]
    L.scilmm_inverse_traces.argtypes = [vp, vp]
    L.scilmm_he_moments.argtypes = [vp, i32, i32, P(dbl), P(dbl)]
    L.scilmm_dist_init.argtypes = [vp, i32, i32, vp, vp, vp]
    L.scilmm_factor_sizes.argtypes = [vp, P(i64), P(i64), P(i64)]
    L.scilmm_dist_work_size.argtypes = [vp, P(i64)]
    L.scilmm_dist_set_work.argtypes = [vp, vp]
    L.scilmm_dist_layout.argtypes = [vp, i32, i32, vp, vp, vp]
    L.scilmm_factor_create_external.argtypes = [vp, vp, vp, vp, P(vp)]
    L.scilmm_dominance.argtypes = [i32, vp, vp, vp, vp, vp]
    L.scilmm_dominance_dev.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.scilmm_dominance_error.restype = C.c_char_p
    L.scilmm_csr_spmm_dev.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp]
    L.scilmm_csr_spmm_error.restype = C.c_char_p
    L.scilmm_order.argtypes = [i32, vp, vp, i32, vp]
    L.scilmm_fill_count.argtypes = [i32, vp, vp, vp, P(i64), P(dbl), P(i32)]
    _lib = L
    return L


def check(status, sym=None, bad_col=None):
    if status == OK:
        return
    if status == ERR_NOT_PD:
        raise NotPositiveDefiniteError(-1 if bad_col is None else int(bad_col))
    msg = {ERR_ARG: "invalid argument", ERR_DEVICE: "HIP device error", ERR_STATE: "invalid call order"}.get(
        status, "error %d" % status)
    if sym:
        detail = lib().scilmm_symbolic_error(sym)
        if detail:
            msg += ": " + detail.decode()
    raise ScilmmError(msg)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_GET_DTYPES = {"sn_rowptr": np.int64, "sn_loff": np.int64, "asm_dst": np.int64, "diag_dst": np.int64,
               "upd_ptr": np.int64, "tile_base": np.int64, "combo_ptr": np.int64, "level_tile_ptr": np.int64,
               "level_pair_ptr": np.int64, "child_ptr": np.int64, "tail_blk_ptr": np.int64, "pat_colptr": np.int64, "inv_off": np.int64}


def symbolic_get(sym, name):
    n = C.c_int64(0)
    check(lib().scilmm_symbolic_get(sym, name.encode(), None, C.byref(n)), sym)
    out = np.empty(n.value, dtype=np.int64 if name.startswith("val_") else _GET_DTYPES.get(name, np.int32))
    check(lib().scilmm_symbolic_get(sym, name.encode(), ptr(out), C.byref(n)), sym)
    return out


COMM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32)

ORDER_METHODS = {"amd": 0, "nesdis": 1, "nesdis_always": 2}


def order(A, method="amd"):
    """Fill-reducing permutation (perm[new] = old) of the symmetric pattern of scipy CSR matrix A."""
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    perm = np.empty(A.shape[0], dtype=np.int32)
    check(lib().scilmm_order(A.shape[0], ptr(indptr), ptr(indices), ORDER_METHODS[method], ptr(perm)))
    return perm


def fill_count(A, perm=None):
    """(nnz(L), sum colcount^2, max colcount) of the factor of A[perm][:, perm]'s pattern."""
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    nz, fl, mx = C.c_int64(0), C.c_double(0), C.c_int32(0)
    p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
    check(lib().scilmm_fill_count(A.shape[0], ptr(indptr), ptr(indices), None if p is None else ptr(p), C.byref(nz),
                                  C.byref(fl), C.byref(mx)))
    return nz.value, fl.value, mx.value


def read_matrix_market(path):
    """scipy.io.mmread(path).tocsr() through the native streaming parser (csrc/mmio.cpp)."""
    import scipy.sparse as sp
    h = C.c_void_p()
    nr, nc, nnz = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    st = lib().scilmm_mm_read(os.fsencode(path), C.byref(h), C.byref(nr), C.byref(nc), C.byref(nnz))
    try:
        if st != OK:
            raise ScilmmError("MatrixMarket: " + (lib().scilmm_mm_error(h) or b"read failed").decode())
        indptr = np.empty(nr.value + 1, dtype=np.int64)
        indices = np.empty(nnz.value, dtype=np.int32)
        data = np.empty(nnz.value, dtype=np.float64)
        check(lib().scilmm_mm_export(h, ptr(indptr), ptr(indices), ptr(data)))
    finally:
        lib().scilmm_mm_free(h)
    return sp.csr_matrix((data, indices, indptr), shape=(nr.value, nc.value))


def dominance(A, parents):
    """Dominance relationship matrix on the pattern of the IBD matrix A (scipy CSR, both halves), built on the device
    (include/scilmm_hip.h scilmm_dominance).  parents: (n, 2) int array, -1 = unknown.  Returns a CSR matrix that shares
    A's pattern.  No CPU fallback."""
    import scipy.sparse as sp
    A = A.tocsr()
    if not A.has_sorted_indices:
        A = A.sorted_indices()
    A.eliminate_zeros()
    n = A.shape[0]
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    data = np.ascontiguousarray(A.data, dtype=np.float64)
    par = np.ascontiguousarray(np.asarray(parents).reshape(n, 2), dtype=np.int32)
    out = np.empty_like(data)
    st = lib().scilmm_dominance(n, ptr(indptr), ptr(indices), ptr(data), ptr(par), ptr(out))
    if st != OK:
        raise ScilmmError("scilmm_dominance failed (%d): %s" % (st, (lib().scilmm_dominance_error() or b"").decode()))
    return sp.csr_matrix((out, A.indices.copy(), A.indptr.copy()), shape=A.shape)


class DeviceCSR(object):
    """A scipy CSR matrix resident in HBM (torch tensors: device memory only) with ``dot(dX)`` = A X for device blocks through
    ``scilmm_csr_spmm_dev`` -- no symbolic analysis involved.  Raises when the GPU cannot be reached: there is no CPU form."""

    def __init__(self, A, torch):
        A = A.tocsr()
        self.torch, self.n = torch, A.shape[0]
        if A.shape[0] != A.shape[1]:
            raise ValueError("square matrix expected")
        self.indptr = torch.from_numpy(np.ascontiguousarray(A.indptr, dtype=np.int64)).cuda()
        self.indices = torch.from_numpy(np.ascontiguousarray(A.indices, dtype=np.int32)).cuda()
        self.data = torch.from_numpy(np.ascontiguousarray(A.data, dtype=np.float64)).cuda()

    def dot(self, dX):
        torch = self.torch
        dX = dX.contiguous()
        if dX.dtype != torch.float64 or dX.dim() != 2 or dX.shape[0] != self.n:
            raise ValueError("an n x r float64 device block is expected")
        dY = torch.empty_like(dX)
        vp = C.c_void_p
        st = lib().scilmm_csr_spmm_dev(self.n, vp(self.indptr.data_ptr()), vp(self.indices.data_ptr()), vp(self.data.data_ptr()),
                                       vp(dX.data_ptr()), dX.shape[1], vp(dY.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
        if st != OK:
            raise ScilmmError("scilmm_csr_spmm_dev failed (%d): %s" % (st, (lib().scilmm_csr_spmm_error() or b"").decode()))
        return dY
