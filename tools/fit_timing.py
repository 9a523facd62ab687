"""Whole REML fit on a BASELINE config through the drop-in Python surface (SURVEY 8d / BASELINE.md section 3: whole-fit
wall-clock, evaluation count, sigma2, beta, std-errs, final nll) -- not the bench metric.

    python tools/fit_timing.py 100k|1m|300k|10k [--out FILE.json] [--cpu-port [--out ...]] [--compare CPU.json]

default        the fit through scilmm_amd.REML on the HIP engine (needs the GPU);
--cpu-port     the SAME fit (same seed, same permutation P, same np.random stream) with every factorization done by
               the BLAS-3 CPU port (oracle/supernodal_cpu.c) through oracle/reml_oracle.fit -- the oracle's own
               restatement of the reference's REML; needs no GPU (test infrastructure, hours at 1M: meant for <= 100k);
--compare F    after the HIP fit, compare sigma2 / beta / std with the CPU-port record F and store the relative
               differences (north_star: sigma2 within 1e-6 relative).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

SEED_FIT = 1


def fit_hip(name, A, C, y, aireml=False, exact_trace=False, front_bits=64):
    P = importlib.import_module("scilmm_amd.SparseCholesky")
    chol = P.SparseCholesky(exact_trace=exact_trace, front_bits=front_bits)
    log, trace = [], []
    orig = P.bolt_gradient_estimation

    def rec(x, *a, **k):
        t = time.time()
        out = orig(x, *a, **k)
        log.append(time.time() - t)
        print("evaluation %d: %.1f s, nll %.10g" % (len(log), log[-1], float(out[0])), flush=True)  # (progress: a 1M evaluation with the selected inverse takes 100 s)
        s2 = np.exp(np.asarray(x)) if k.get("take_exp", True) else np.asarray(x)  # (AI-REML evaluates at sigma2 itself)
        trace.append((s2.tolist(), float(out[0]), np.asarray(out[1]).tolist()))
        return out

    P.bolt_gradient_estimation = rec
    # where the time OUTSIDE the evaluations goes: wall-clock of the named steps, nested calls counted once
    phases, depth = {}, [0]

    def timed(name, fn):
        def wrap(*a, **k):
            outer = depth[0] == 0 and not (log and False)
            depth[0] += 1
            t = time.time()
            try:
                return fn(*a, **k)
            finally:
                depth[0] -= 1
                if outer:
                    phases[name] = phases.get(name, 0.0) + time.time() - t
        return wrap

    saved = {}
    for name in ("HE", "_final_factor", "_compute_hess_device", "compute_varcomp_stderr", "estimate_fixed_effects"):
        saved[name] = getattr(P, name)
        setattr(P, name, timed(name, saved[name]))
    saved_init = P._DeviceProjector.__init__
    P._DeviceProjector.__init__ = timed("V^-1 [C | y] sweep of the post-fit algebra", saved_init)
    saved_eng = P.SparseCholesky.engine_for
    eng_t = [0.0]

    def eng_wrap(self, mats):
        t = time.time()
        try:
            return saved_eng(self, mats)
        finally:
            eng_t[0] += time.time() - t

    P.SparseCholesky.engine_for = eng_wrap
    np.random.seed(SEED_FIT)
    t0 = time.time()
    res = P.REML(chol, [A], C, y, aireml=aireml)
    tot = time.time() - t0
    P.bolt_gradient_estimation = orig
    for name, fn in saved.items():
        setattr(P, name, fn)
    P._DeviceProjector.__init__ = saved_init
    P.SparseCholesky.engine_for = saved_eng
    phases["engine_for (analysis + value upload on the first call, checksums later; partly INSIDE evaluations)"] = eng_t[0]
    best = min(trace, key=lambda t: t[1])
    return {"engine": "HIP (scilmm_amd.REML, fused evaluation, device-resident n x 100 blocks)",
            "optimiser": "AI-REML" if aireml else "L-BFGS-B (the reference's)",
            "front_bits": front_bits,
            "trace": "exact (selected inverse)" if exact_trace else "Monte-Carlo, 100 vectors (the reference's)",
            "fit_wall_s": tot, "evaluations": len(log), "first_evaluation_s": log[0],
            "median_later_evaluation_s": float(np.median(log[1:])) if len(log) > 1 else None,
            "outside_evaluations_s": tot - sum(log), "phases_s": phases,
            "sigma2": np.asarray(res["covariance coefficients"]).tolist(),
            "beta": np.asarray(res["covariates coefficients"]).tolist(),
            "std": np.asarray(res["covariance std"]).tolist(),
            "final_nll": trace[-1][1], "lowest_nll": best[1], "trajectory": trace}


def fit_cpu_port(name, A, C, y):
    from oracle import oracle as O
    from oracle import reml_oracle as RO
    from scilmm_amd.factor import Symbolic
    n = A.shape[0]
    sym = Symbolic([A, sp.identity(n, format="csr")], upload=False)  # the engine's analysis (host only): same P
    arrays, colptr = sym.arrays(), sym.get("pat_colptr")
    trace, log = [], []
    t_last = [time.time()]

    def factor_of(V):
        return O.CPUPortFactor(arrays, colptr, V)

    np.random.seed(SEED_FIT)
    t0 = time.time()
    tr = []
    s2, beta, std = RO.fit([A], C, y, reml=True, sim_num=100, factor_of=factor_of, trace=tr)
    tot = time.time() - t0
    for x, nll, g in tr:
        trace.append((np.exp(x).tolist(), float(nll), np.asarray(g).tolist()))
    return {"engine": "CPU port (oracle/reml_oracle.fit + oracle/supernodal_cpu.c, %d threads)" % bench._effective_cpus(),
            "fit_wall_s": tot, "evaluations": len(trace), "sigma2": s2.tolist(), "beta": beta.tolist(), "std": std.tolist(),
            "final_nll": trace[-1][1], "lowest_nll": min(t[1] for t in trace), "trajectory": trace}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="100k", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--cpu-port", action="store_true")
    ap.add_argument("--aireml", action="store_true", help="average-information iteration instead of L-BFGS-B")
    ap.add_argument("--exact-trace", action="store_true", help="tr(V^-1 A_k) from the selected inverse instead of the Monte-Carlo estimate")
    ap.add_argument("--front-bits", type=int, default=64, choices=[32, 64], help="32: fp32-product fronts (BASELINE configs[4]'s arithmetic)")
    ap.add_argument("--compare", default=None)
    args = ap.parse_args()
    t0 = time.time()
    A, C, y = bench.build_problem(args.workload, 0)
    n = A.shape[0]
    rec = {"workload": args.workload, "n": int(n), "nnz_A": int(A.nnz), "generate_s": time.time() - t0,
           "seed_pedigree": 0, "seed_fit": SEED_FIT, "sim_num": 100,
           "reference": "REML(SparseCholesky(), [A], cov, y) -- /root/reference/scilmm/SparseCholesky.py:177-189"}
    print("problem: n=%d nnz=%d  (%.1f s)" % (n, A.nnz, rec["generate_s"]), flush=True)
    rec.update(fit_cpu_port(args.workload, A, C, y) if args.cpu_port else fit_hip(args.workload, A, C, y, aireml=args.aireml, exact_trace=args.exact_trace, front_bits=args.front_bits))
    if args.compare:
        other = json.load(open(args.compare))
        rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())
        rec["vs_cpu_port"] = {"file": args.compare, "sigma2_rel": rel(rec["sigma2"], other["sigma2"]),
                              "beta_rel": rel(rec["beta"], other["beta"]), "std_rel": rel(rec["std"], other["std"]),
                              "evaluations_cpu": other["evaluations"],
                              "shared_evaluations_nll_rel": max(abs(a[1] - b[1]) / abs(b[1]) for a, b in
                                                                zip(rec["trajectory"], other["trajectory"]))}
        print("vs CPU port:", rec["vs_cpu_port"], flush=True)
    print("fit: %.1f s, %d evaluations, sigma2 %s beta %s std %s final nll %.10g" %
          (rec["fit_wall_s"], rec["evaluations"], rec["sigma2"], rec["beta"], rec["std"], rec["final_nll"]), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
