"""Time scilmm_spmm_dev (Y = A_k X on the device) on a BASELINE workload.  usage: python tools/spmm_timing.py 300k [r]"""
import ctypes, os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from scilmm_amd.factor import Symbolic
name = sys.argv[1] if len(sys.argv) > 1 else "100k"
r = int(sys.argv[2]) if len(sys.argv) > 2 else 103
A, C, y = bench.build_problem(name, 0)
n = A.shape[0]
sym = Symbolic([A, sp.identity(n, format="csr")])
X = torch.randn(n, r, dtype=torch.float64, device="cuda")
Y = torch.empty_like(X)
vp = ctypes.c_void_p
for k in (0, 1):
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        sym.spmm_dev(k, vp(X.data_ptr()), r, vp(Y.data_ptr())); sym.sync()
        dt = time.time() - t
    ref = (A if k == 0 else sp.identity(n)).dot(X[:, :3].cpu().numpy())
    print("matrix %d: %.3f s for %d columns, nnz %d, check %.2e" % (k, dt, r, A.nnz if k == 0 else n, np.abs(Y[:, :3].cpu().numpy() - ref).max()), flush=True)
