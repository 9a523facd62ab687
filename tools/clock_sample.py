"""Sample the GPU clocks (rocm-smi) while a loop of solves / factorizations runs: are the sweeps clock-starved?"""
import sys, ctypes, time, subprocess, threading
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
import bench
from scilmm_amd.factor import Symbolic
A, C, y = bench.build_problem("100k", 0)
n = A.shape[0]
sym = Symbolic([A, sp.identity(n, format="csr")])
fac = sym.factorize([0.4, 0.6])
dev = torch.device("cuda", 0)
B = torch.randn(n, 103, dtype=torch.float64, device=dev)
X = torch.empty_like(B)
stop = False
samples = []
def sampler(tag):
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            s = [l for l in out.splitlines() if "sclk" in l.lower()]
            samples.append((tag, s[0].strip() if s else out[:80]))
        except Exception as e:
            samples.append((tag, repr(e)))
        time.sleep(0.2)
for tag, fn in (("solve loop", lambda: (fac.solve_dev(ctypes.c_void_p(B.data_ptr()), 103, ctypes.c_void_p(X.data_ptr())), sym.sync())),
                ("factorize loop", lambda: fac.refactorize([0.4, 0.6]))):
    stop = False
    th = threading.Thread(target=sampler, args=(tag,)); th.start()
    t0 = time.time()
    while time.time() - t0 < 4.0:
        fn()
    stop = True; th.join()
for s in samples[:6] + samples[-6:]:
    print(s)
