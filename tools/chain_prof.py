"""Anatomy of a forward chain-sweep step (debug build: -DSCILMM_CHAIN_PROF, library passed via SCILMM_HIP_LIB)."""
import ctypes as C, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
import bench
from scilmm_amd import _lib
from scilmm_amd.factor import Symbolic
A, Cc, y = bench.build_problem("100k", 0)
n = A.shape[0]
sym = Symbolic([A, sp.identity(n, format="csr")])
f = sym.factorize([0.4, 0.6])
B = np.random.default_rng(0).standard_normal((n, 103))
f(B); f(B)
lib = _lib.lib()
N = 200
buf = (C.c_ulonglong * (8 * N))()
lib.scilmm_debug_chain_prof(buf, N)
t = np.array(list(buf), dtype=np.float64).reshape(N, 8) / 100.0  # us
t = t[t[:, 5] > 0]
T = len(t)
s = slice(T - 100, T)
prev5 = t[T - 101:T - 1, 5]
names = ["prev all-waves-drained -> flag observed", "flag observed -> x staged", "x staged -> last pair done",
         "last pair done -> w_i in LDS", "w_i in LDS -> wave 0 stores issued", "wave 0 stores -> all waves drained"]
vals = [t[s, 0] - prev5, t[s, 1] - t[s, 0], t[s, 2] - t[s, 1], t[s, 3] - t[s, 2], t[s, 4] - t[s, 3], t[s, 5] - t[s, 4]]
print("fronts", T, " mean step %.1f us" % np.diff(t[s, 5]).mean())
for nm, v in zip(names, vals):
    print("  %-42s %.2f us" % (nm, v.mean()))
