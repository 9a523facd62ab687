#!/usr/bin/env python
"""ORACLE / TEST INFRASTRUCTURE -- a value-level digest of the 300k probe workload computed by the BLAS-3 CPU port
(oracle/supernodal_cpu.c), stored as a fixture-sized file that the GPU test `test_300k_values_against_cpu_port_digest`
asserts (above 100k the parity tests were property-only).

    python oracle/make_digest_300k.py [--out tests/golden/D1_300k_cpu_port_digest.npz]

Run ONCE in the build container (no GPU: the analysis is host code, the factorization the CPU port; ~15 minutes on 8 cores).
What is stored: log det V, and of X = V^-1 B for three seeded columns the entries of every 997th row, the column sums and
the column 2-norms -- V = 0.4 A + 0.6 I of `bench.build_problem("300k", seed 0)`.  log det and V^-1 B do not depend on the
permutation, so the GPU side may order the matrix as it likes.
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STRIDE = 997
SIGMA2 = (0.4, 0.6)


def rhs(n):
    return np.random.default_rng(300).standard_normal((n, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "D1_300k_cpu_port_digest.npz"))
    ap.add_argument("--workload", default="300k")
    args = ap.parse_args()
    import bench
    from oracle import oracle as O
    from scilmm_amd.factor import Symbolic
    t0 = time.time()
    A, C, y = bench.build_problem(args.workload, 0)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    print("problem: n=%d nnz=%d (%.0f s)" % (n, A.nnz, time.time() - t0), flush=True)
    t0 = time.time()
    sym = Symbolic([A, I], upload=False, max_width=512)   # (512-column blocks: what a CPU supernodal code wants)
    arrays, colptr = sym.arrays(), sym.get("pat_colptr")
    print("analysis: %.0f s, nnz(L) %.3g, %.3g flops" % (time.time() - t0, sym.info().nnzL, sym.info().flops), flush=True)
    t0 = time.time()
    f = O.CPUPortFactor(arrays, colptr, (SIGMA2[0] * A + SIGMA2[1] * I).tocsr())
    print("CPU port factorization: %.0f s" % (time.time() - t0), flush=True)
    B = rhs(n)
    X = f(B)
    V = (SIGMA2[0] * A + SIGMA2[1] * I).tocsr()
    resid = float(np.abs(V @ X - B).max() / np.abs(B).max())
    print("residual of the CPU port's solve: %.2e" % resid, flush=True)
    assert resid < 1e-11
    np.savez(args.out, workload=args.workload, n=n, nnz_A=A.nnz, sigma2=np.array(SIGMA2), logdet=f.logdet(), stride=STRIDE,
             X_rows=X[::STRIDE].copy(), X_sum=X.sum(axis=0), X_norm=np.sqrt((X * X).sum(axis=0)), residual=resid,
             a_checksum=float(A.data.sum()), nnzL=int(sym.info().nnzL))
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
