"""Timing-only runs of the factorization with diagnostic ablations (SCILMM_ABLATE); results are NOT valid numbers.
Needs a diagnostic build (make -C scilmm_amd/csrc DIAG=1) and SCILMM_TUNING=1.
usage: SCILMM_ABLATE=3 SCILMM_NO_LOOKAHEAD=1 python tools/ablate_timing.py"""
import ctypes as C
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
import bench
from scilmm_amd import _lib
from scilmm_amd.factor import Symbolic

A, Cc, y = bench.build_problem("100k", 0)
n = A.shape[0]
sym = Symbolic([A, sp.identity(n, format="csr")])
sym.set_profiling(True)
lib = _lib.lib()
h = C.c_void_p(); bad = C.c_int32(-1)
s2 = np.array([0.4, 0.6])
rc = lib.scilmm_factorize(sym._h, s2.ctypes.data_as(C.c_void_p), C.byref(h), C.byref(bad))
for it in range(2):
    rc = lib.scilmm_refactorize(h, s2.ctypes.data_as(C.c_void_p), C.byref(bad))
t = sym.timing()
print("rc", rc, {k: round(v, 1) for k, v in t.items() if k in ("factor_ms", "update_ms", "potrf_ms", "trsm_ms", "reduce_cells_ms", "update_union_ms")})
