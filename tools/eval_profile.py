import sys, time, importlib, cProfile, pstats, io
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
import bench
P = importlib.import_module("scilmm_amd.SparseCholesky")
A, C, y = bench.build_problem("100k", 0)
chol = P.SparseCholesky()
x = np.log(np.array([0.4, 0.6]))
np.random.seed(1)
for _ in range(2):
    P.bolt_gradient_estimation(x, chol, [A, sp.identity(A.shape[0], format="csr")], C, y, True, 100, False)
pr = cProfile.Profile(); pr.enable()
t = time.time()
for _ in range(3):
    P.bolt_gradient_estimation(x, chol, [A, sp.identity(A.shape[0], format="csr")], C, y, True, 100, False)
dt = (time.time() - t) / 3
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print("per evaluation %.3f s" % dt)
print("\n".join(s.getvalue().splitlines()[:45]))
