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
