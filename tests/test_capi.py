"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/scilmm_hip.h declares, and
fails loudly (no CPU fallback) when a numeric call is made without a GPU."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from scilmm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "scilmm_hip.h")).read()
    declared = set(re.findall(r"\b(scilmm_[a-z_A-Z0-9]+)\s*\(", header))
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.SYMBOLS)
    assert b"gfx950" in L.scilmm_version()


def test_numeric_calls_fail_loudly_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    from scilmm_amd.factor import Symbolic
    A = sp.identity(4, format="csr") * 2.0
    with pytest.raises(_lib.ScilmmError):
        Symbolic([A])  # values_upload needs the device
    sym = Symbolic([A], upload=False)
    with pytest.raises(_lib.ScilmmError):
        sym.factorize([1.0])


def test_drop_in_surface_names():
    import scilmm_amd
    for name in ["SparseCholesky", "REML", "HE", "run_estimates", "run_estimates_from_paths", "bolt_gradient_estimation",
                 "estimate_var_comps", "compute_hess", "compute_varcomp_stderr", "matrices_weighted_sum"]:
        assert hasattr(scilmm_amd, name)
    from scilmm_amd.Estimation.LMM import LMM, SparseCholesky  # noqa: F401  (reference SciLMM.py:7)


def test_bad_arguments_are_rejected():
    from scilmm_amd.factor import Symbolic
    with pytest.raises(ValueError):
        Symbolic([sp.identity(3, format="csr"), sp.identity(4, format="csr")], upload=False)
    with pytest.raises(ValueError):
        Symbolic([sp.identity(3, format="csr")], perm=np.arange(4), upload=False)
    # the assembly maps may only be released once the values are in HBM (the device plan reads them)
    from scilmm_amd._lib import ScilmmError
    sym = Symbolic([sp.identity(3, format="csr")], upload=False)
    with pytest.raises(ScilmmError):
        sym.release_host_maps()


def test_round4_entry_points_validate_and_never_fall_back(gpu_available):
    """The entry points added in round 4 reject bad arguments on the host and -- without a GPU -- fail with the device error
    instead of computing anything on the CPU: `scilmm_csr_spmm_dev`, `scilmm_dominance_values_device`, the value-less
    `PatternCSR` analysis, `scilmm_timing.n_late_split`."""
    import ctypes as C
    from scilmm_amd.factor import PatternCSR, Symbolic
    L = _lib.lib()
    vp = C.c_void_p
    one = vp(8)   # (a non-null dummy: the argument checks come before any dereference)
    assert L.scilmm_csr_spmm_dev(-1, one, one, one, one, 1, vp(16), None) == _lib.ERR_ARG
    assert L.scilmm_csr_spmm_dev(4, one, one, one, one, 0, vp(16), None) == _lib.ERR_ARG      # r <= 0
    assert L.scilmm_csr_spmm_dev(4, one, one, one, one, 3, one, None) == _lib.ERR_ARG         # X and Y alias
    assert L.scilmm_csr_spmm_dev(4, None, one, one, one, 3, vp(16), None) == _lib.ERR_ARG
    assert L.scilmm_csr_spmm_dev(0, one, None, None, one, 3, vp(16), None) == _lib.OK         # empty matrix: nothing to do
    # a value-less pattern analyses like the matrix it came from
    A = (sp.random(60, 60, density=0.1, random_state=3, format="csr") + sp.identity(60, format="csr")).tocsr()
    A = (A + A.T).tocsr()
    A.sort_indices()
    P = PatternCSR(A.indptr, A.indices, 60)
    I = sp.identity(60, format="csr")
    sp_sym, sa_sym = Symbolic([P, I], upload=False), Symbolic([A, I], upload=False)
    assert np.array_equal(sp_sym.P(), sa_sym.P()) and sp_sym.info().nnzL == sa_sym.info().nnzL
    with pytest.raises(ValueError):
        PatternCSR(A.indptr[:-1], A.indices, 60)
    par = np.full((60, 2), -1, dtype=np.int32)
    for bad in ((0, 0), (0, 5), (-1, 0), (1, 0)):   # k_dst == k_src, out of range, the diagonal-only identity as a target
        assert L.scilmm_dominance_values_device(sp_sym._h, bad[0], bad[1], 60, _lib.ptr(par)) == _lib.ERR_ARG
    assert L.scilmm_dominance_values_device(sp_sym._h, 0, 0, 59, _lib.ptr(par)) == _lib.ERR_ARG
    assert "n_late_split" in dict(_lib.Timing._fields_)
    if not gpu_available:
        sym3 = Symbolic([P, P, I], upload=False)
        with pytest.raises(_lib.ScilmmError, match="HIP|device"):
            sym3.dominance_values_from(1, 0, par)            # needs the device: no CPU form exists
        with pytest.raises(_lib.ScilmmError, match="HIP|device"):
            sym3.ibd_values_from_pedigree(0, par)
