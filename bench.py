#!/usr/bin/env python
"""Headline benchmark: per-REML-evaluation factorize -> solve -> log-det of V = s2_g A + s2_e I on a simulated
pedigree (BASELINE.json metric "REML factorize+solve wall-clock (s) and nnz(L)/s, 1M-individual pedigree").

    python bench.py --gpus N --steps K --warmup W [--workload 1m|100k|300k|10k] [--budget-s S]

A step = one pass of the hot path over one cohort: device assembly of V, numeric supernodal Cholesky,
log-det, and ONE fused solve with r = c + 1 + s = 103 right-hand sides ([C | y | Z]), everything
resident in HBM when the timed region starts.  value = nnz(L) processed by all ranks / wall time.

The default workload is BASELINE configs[2] -- the 1M-individual pedigree the metric is quoted on (n = 828k after
the unrelated-drop, nnz(L) = 1.5e10 = 123 GB, 1.6 PFLOP per factorization, ~40 s per step on one MI355X).  A driver
that asks for --steps 20 --warmup 5 cannot get 25 such steps inside its time limit, so the step counts are BUDGETED
by wall clock (--budget-s, default 450 s for the whole process): the first evaluation (which also builds the
device plan) is the warm-up, then as many timed steps as fit are run (at least one); the line reports the counts
actually run as `steps` / `warmup` and the requested ones as `steps_requested` / `warmup_requested`.  Small
workloads (--workload 100k) fit the requested counts and run them unchanged.

For N > 1 the N ranks factorize the SAME ONE cohort together (strong scaling, BASELINE configs[3]): the panels of the
dense tail of the block elimination tree -- > 99.9 % of the flops and of the storage at 1M -- are owned 1-D
block-cyclically; a finished panel is broadcast by its owner over RCCL/xGMI into the receivers' ring, applied to
their own targets and dropped (fan-out, rank-local storage: scilmm_amd/dist.py), and the 103-column solve runs
collectively (one small all-reduce / broadcast per tail block).  Rank 0 prints one JSON line.  `python bench.py
--gpus N` starts its N ranks itself; under a launcher (RANK / WORLD_SIZE set) it is one of them.
"""
import argparse
import ctypes
import json
import os
import sys
import time

T_PROCESS_START = time.time()

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n individuals, sparsity_factor)  -- BASELINE.json configs[0..2]
    "10k": (10000, 0.001),
    "100k": (100000, 0.005),
    "300k": (300000, 0.003),     # scaling probe between configs[1] and configs[2] (2.0e9 nnz(L), 7.6e13 flops)
    "1m": (1000000, 0.001),      # configs[2]: n_eff 828k, nnz(L) 1.5e10 (123 GB), 1.6e15 flops -- see DESIGN.md
    # configs[4]: 3M individuals, A + D + I, 8 ranks, fp32-product fronts (sparsity_factor 3e-4, 2.7e9 entries of A).
    # Needs a multi-GPU node (~1.1 TB of factor: DESIGN.md section 7) and ~150 GB of host memory on rank 0; launched by hand:
    #   python bench.py --gpus 8 --workload 3m --components A,D --front-bits 32 --budget-s 3000
    "3m": (3000000, 0.0003),
}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_16x16x4_f32), 155 measured
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix; MI355X_MICROARCH.md lists no fp64 figure
HBM_PEAK_GBS = 8000.0
CPU_BASELINE_WORKLOAD = "100k"   # bounded sample for the host-cores baseline (a 1.6 PFLOP CPU run would take hours)
CPU_BASELINE_RESERVE_S = 75.0


def build_problem(name, seed, components="A", parents=False):
    """components "A": (A, C, y); "A,D": ([A, D], C, y) with the dominance matrix D built on the device from A and the
    parent table (BASELINE configs[4]'s second variance component; reference scilmm/Matrices/Dominance.py:12-43).
    parents=True: also the parent table of the cohort (after the unrelated-drop), from which the ranks of a multi-GPU run
    build the VALUES of A and D in their own HBM."""
    from scilmm_amd.harness.pedigree import make_problem
    n, sf = WORKLOADS[name]
    out = make_problem(n, sf, seed=seed, with_dominance="device" if components != "A" else False, return_parents=parents)
    mats = out[0]
    return (mats[0] if components == "A" else mats,) + tuple(out[1:])


def _effective_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


def cpu_baseline(A, r, sample_name):
    """Supernodal BLAS-3 LL^T + solve on the host cores (oracle/supernodal_cpu.c).  Same ordering algorithm as
    the GPU run but its own analysis with 512-column blocks (what a CPU supernodal code wants)."""
    from oracle import oracle as O
    from scilmm_amd.factor import Symbolic
    n = A.shape[0]
    # BLAS threads = the CPUs this process may really use (affinity mask AND cgroup quota: the GPU box shows 256 logical
    # CPUs but grants 16 CPUs' worth of time; 64 BLAS threads under that quota ran 17-29 s, throttled at random)
    cpus = _effective_cpus()
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threadpool_limits(limits=cpus, user_api="blas")
        threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        threads = cpus
    csym = Symbolic([A, sp.identity(n, format="csr")], upload=False, max_width=512)
    cpu = O.SupernodalCPU(csym.arrays(), n)
    # values of V = 0.4 A + 0.6 I in pattern-slot order = CSC order of tril(V[P][:,P]) (diagonal first)
    perm = csym.get("perm")
    Lw = sp.tril(A.tocsr()[perm][:, perm]).tocsc()
    Lw.sort_indices()
    assert np.array_equal(Lw.indptr, csym.get("pat_colptr"))
    vals = 0.4 * Lw.data
    vals[Lw.indptr[:-1]] += 0.6
    t0 = time.time()
    cpu.assemble(vals)
    cpu.factorize()
    t_fact = time.time() - t0
    rng = np.random.default_rng(0)
    Y = np.asfortranarray(rng.standard_normal((n, r)))
    t0 = time.time()
    cpu.solve_permuted(Y)
    t_solve = time.time() - t0
    cinfo = csym.info()
    gpu = None
    try:
        # the SAME sample on the GPU (3 steps after one warm-up), so that the two nnz(L)/s figures are comparable: nnz(L)/s is
        # not comparable across workloads (the 1M factor costs 10x more flops per nonzero than the 100k one)
        import ctypes
        import torch
        gsym = Symbolic([A, sp.identity(n, format="csr")])
        dBg = torch.from_numpy(np.ascontiguousarray(rng.standard_normal((n, r)))).cuda()
        dXg = torch.empty_like(dBg)
        gfac = gsym.factorize([0.4, 0.6])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gfac.refactorize([0.4, 0.6])
            gfac.logdet()
            gfac.solve_dev(ctypes.c_void_p(dBg.data_ptr()), r, ctypes.c_void_p(dXg.data_ptr()))
            gsym.sync()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 3
        ginfo = gsym.info()
        gpu = {"value": ginfo.nnzL / tg, "unit": "nnz(L)/s", "seconds_per_step": tg, "nnzL": int(ginfo.nnzL),
               "tflops": ginfo.flops / tg / 1e12, "note": "same cohort, same r, this GPU, 128-column blocks"}
        del gfac, gsym, dBg, dXg
    except Exception as e:
        gpu = {"value": None, "note": "failed: %r" % (e,)}
    return {"value": cinfo.nnzL / (t_fact + t_solve), "unit": "nnz(L)/s", "cores": int(threads), "kind": "port",
            "baseline_is": "this repository's own CPU port (oracle/supernodal_cpu.c); no reference CPU path was timed: the "
                           "reference's scikit-sparse / CHOLMOD is not available on this box",
            "gpu_same_sample": gpu, "tflops": cinfo.flops / t_fact / 1e12,
            "sample": "the %s cohort (n=%d, nnz(L)=%.3g, %.3g flops), full factorization once: supernodal LL^T (%.2f s) + "
                      "%d-column solve (%.2f s); oracle/supernodal_cpu.c with SciPy-bundled OpenBLAS (%d BLAS threads = the CPU quota; %d "
                      "logical CPUs visible), own AMD ordering, 512-column supernode blocks; CHOLMOD unavailable on this box"
                      % (sample_name, n, cinfo.nnzL, cinfo.flops, t_fact, r, t_solve, threads, len(os.sched_getaffinity(0))),
            "sample_workload": sample_name, "sample_nnzL": int(cinfo.nnzL), "sample_flops": cinfo.flops,
            "factor_s": t_fact, "solve_s": t_solve, "flops_per_s": cinfo.flops / t_fact, "logdet": cpu.logdet()}


def plan_only(args):
    """`--plan-only`: what ONE rank of `--gpus N` holds in HBM for a workload, from the REAL analysis of its pattern (VERDICT r3
    item 1d).  Host part (no GPU): the pedigree's pattern only (no matrix value exists anywhere), the symbolic analysis, the
    distribution rule read from the library (`scilmm_dist_layout`) -> exact byte counts of everything whose size the analysis
    fixes.  Device part (when a GPU is present): the engine of rank `--plan-rank` is built WITHOUT peers -- its device plan
    (work items, descriptors, cell lists), its share of the factor, the values of A (and D) computed in HBM -- no
    factorization is run, and the HBM in use is read from the runtime.  Prints one JSON object; returns the exit status."""
    import resource
    from scilmm_amd.dist import tail_layout
    from scilmm_amd.factor import Symbolic
    from scilmm_amd.harness.pedigree import make_pattern_problem
    world, rank = args.gpus, args.plan_rank
    if world < 1 or not 0 <= rank < world:
        sys.stderr.write("--plan-only: need --gpus >= 1 and 0 <= --plan-rank < --gpus\n")
        return 2
    k3 = args.components != "A"
    n0, sf = WORKLOADS[args.workload]
    t0 = time.time()
    P, par, _ = make_pattern_problem(n0, sf, seed=int(os.environ.get("SCILMM_BENCH_SEED", "0")))
    t_gen = time.time() - t0
    n = P.shape[0]
    mats_e = [P] + ([P] if k3 else []) + [sp.identity(n, format="csr")]
    t0 = time.time()
    have_gpu = False
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        pass
    eng = None
    if have_gpu and world > 1:
        from scilmm_amd.dist import HipChainEngine
        eng = HipChainEngine(mats_e, rank, world, None, "cuda:0")   # no peers: nothing collective is ever issued
        sym = eng.sym
    else:
        sym = Symbolic(mats_e, upload=False)
    t_sym = time.time() - t0
    info = sym.info()
    ns = info.nsuper
    owner, loff, (first, Wg, G) = tail_layout(sym._h, ns, rank, world)
    sn_loff = sym.get("sn_loff")
    sn_rowptr = sym.get("sn_rowptr")
    glob = np.append(sn_loff[:ns], info.nnzL_stored) if sn_loff.size == ns else sn_loff
    sizes = np.diff(glob)
    nT = ns - first
    prelude = int(glob[first]) if first < ns else int(info.nnzL_stored)
    slot = int(((sizes[first:] + 1) // 2 * 2).max()) if nT > 0 else 0
    mine = [j for j in range(nT) if world > 1 and j % world == rank]
    own = int(((sizes[first:][mine] + 1) // 2 * 2).sum()) if world > 1 else int(info.nnzL_stored) - prelude
    ring = (min(G, nT) * slot) if world > 1 else 0
    assert world == 1 or prelude + own + ring == int(loff[-1]), (prelude, own, ring, int(loff[-1]))
    max_m = int(np.diff(sn_rowptr).max())
    inv_doubles = int(sym.get("inv_off")[-1])
    nnzp = int(info.nnz_pattern)
    n_general = 2 if k3 else 1
    r = 103
    GB = 1e-9
    comp = {
        "factor_prelude_replicated": 8.0 * prelude,
        "factor_own_tail_panels": 8.0 * own,
        "factor_ring_slots": 8.0 * ring,
        "factor_slack": 8.0 * (4 * max_m + 4 * 128),
        "inverse_diagonal_blocks_replicated": 8.0 * inv_doubles,
        "fp32_shadow_of_own_tail_and_ring": (4.0 * (own + ring + 16384)) if args.front_bits == 32 else 0.0,
        "values_A_k_in_slot_order": 8.0 * (n_general * nnzp + n),
        "assembly_maps_asm_dst_pat_row_colptr": 8.0 * nnzp + 4.0 * nnzp + 8.0 * 2 * n,
        "sweep_work_buffers_W_X_ACC": 8.0 * (3 if world > 1 else 2) * n * 128,
        "bench_io_B_X_and_refinement_blocks": 8.0 * n * r * (4 if args.front_bits == 32 else 2),
    }
    total_known = sum(comp.values())
    out = {"plan_only": True, "workload": args.workload, "components": args.components, "front_bits": args.front_bits,
           "world": world, "rank": rank, "n": n, "nnz_pattern_tril": nnzp, "nnzL": int(info.nnzL), "nnzL_stored": int(info.nnzL_stored),
           "factor_flops": info.flops, "nsuper": ns, "tail_panels": nT, "tail_columns": int(n - sym.get("sn_start")[first]) if nT else 0,
           "group_size": Wg, "ring_slots": min(G, nT) if world > 1 else 0, "largest_panel_MB": 8e-6 * slot,
           "bytes_fixed_by_the_analysis_GB": {k: v * GB for k, v in comp.items()},
           "sum_fixed_by_the_analysis_GB": total_known * GB,
           "generate_s": t_gen, "analysis_s": t_sym,
           "host_peak_rss_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024 * GB}
    if have_gpu:
        dev = torch.device("cuda", 0)
        t0 = time.time()
        if eng is None:
            sym.upload_values()     # (the identity: the only matrix with host values; builds the device plan)
        sym.ibd_values_from_pedigree(0, par)
        if k3:
            sym.dominance_values_from(1, 0, par)
        torch.cuda.synchronize()
        free_b, total_b = torch.cuda.mem_get_info(dev)
        used = float(total_b - free_b)
        # what the probe holds = everything above except the factor when world == 1 (allocated by the first factorization), the
        # fp32 shadow (allocated by the first fp32-front factorization), the sweep buffers of a single-GPU handle (first solve)
        # and the bench's blocks
        held = dict(comp)
        for k in ("fp32_shadow_of_own_tail_and_ring", "bench_io_B_X_and_refinement_blocks"):
            held[k] = 0.0
        if world == 1:
            for k in ("factor_prelude_replicated", "factor_own_tail_panels", "factor_ring_slots", "factor_slack",
                      "inverse_diagonal_blocks_replicated", "sweep_work_buffers_W_X_ACC"):
                held[k] = 0.0
        plan = used - sum(held.values())
        out.update({"hbm_in_use_by_the_probe_GB": used * GB, "device_plan_and_runtime_GB": plan * GB,
                    "hbm_per_rank_GB": (total_known + plan) * GB, "hbm_total_GB": total_b * GB,
                    "device_plan_s": time.time() - t0,
                    "note": "device_plan_and_runtime = HBM in use after building this rank's plan and values, minus the analysis-fixed "
                            "terms the probe holds (includes the runtime's own ~1 GB and allocator rounding); hbm_per_rank = "
                            "analysis-fixed terms + that"})
    else:
        out["note"] = ("no GPU here: the device plan (work items, descriptors, cell lists: 19 GB at 1M on one GPU, ~1/world of it plus 0.2 GB "
                       "per rank when the tail is distributed) is not included -- run the same command on a GPU box for the measured figure")
    print(json.dumps(out))
    return 0


def spawn_ranks(n):
    """Launcher half of `python bench.py --gpus N`: N children of this interpreter with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set (127.0.0.1, a free port), the same command line; rank 0's stdout carries the JSON line.  The parent makes
    no GPU call at all.  Returns the exit status: 0 only if every rank exited 0."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad:
                # one rank failed: the others would wait in a collective until it times out
                time.sleep(5.0)
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                worst = bad[0]
                break
    finally:
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                p.kill()
    for p in procs:
        if p.returncode not in (0, None) and worst == 0:
            worst = p.returncode
    if worst != 0:
        sys.stderr.write("bench.py: a rank exited with status %s\n" % worst)
    return 1 if worst != 0 else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SCILMM_BENCH_WORKLOAD", "1m"), choices=sorted(WORKLOADS))
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("SCILMM_BENCH_BUDGET_S", "450")),
                    help="wall-clock budget of the whole process; the step counts are cut to fit it (>= 1 timed step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clean-profile", action="store_true",
                    help="skip the extra untimed evaluation that measures the dominant kernel's launches without overlap")
    ap.add_argument("--components", default="A", choices=["A", "A,D"],
                    help="A,D: K = 3 (additive + dominance + identity), BASELINE configs[4]'s model -- NOT the headline metric")
    ap.add_argument("--front-bits", type=int, default=64, choices=[32, 64],
                    help="32: BASELINE configs[4]'s arithmetic (fp32 MFMA fronts, fp64 sums) -- NOT the headline metric; "
                         "the line then says so in `dtype` and `metric`")
    ap.add_argument("--serialised", action="store_true",
                    help="measurement mode: EVERY evaluation queues its look-ahead launches on one side stream (scilmm_set_profiling 2), so "
                         "that a `rocprofv3 --kernel-trace --stats` of this command shows the dominant kernel's launch durations without "
                         "overlap -- the figure `roofline.achieved` quotes; slower as a whole, same results; the line says so")
    ap.add_argument("--no-engine-profiling", action="store_true",
                    help="do not bracket the kernel classes with HIP events (for rocprofv3 --pmc passes: the counters serialise the "
                         "dispatches anyway and ~20k extra event packets per factorization only load the intercepted queues); the "
                         "line's roofline object is then null")
    ap.add_argument("--dump-maps", default=None,
                    help="write /proc/self/maps to this file right before the first evaluation (to symbolise a profiler-side crash offline)")
    ap.add_argument("--plan-only", action="store_true",
                    help="analyse the workload's pattern on the host and print, as one JSON object, what ONE rank of --gpus N holds in "
                         "HBM (exact counts of the analysis and of the distribution rule; no GPU needed).  With a GPU present the "
                         "device plan of rank --plan-rank is built as well (no peers, no factorization) and the HBM in use is measured.")
    ap.add_argument("--plan-rank", type=int, default=0)
    args = ap.parse_args()

    if args.plan_only:
        raise SystemExit(plan_only(args))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It starts the N ranks as
        # fresh children BEFORE anything here touches the GPU (it never does) and leaves with their worst exit code.
        raise SystemExit(spawn_ranks(args.gpus))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to print a line "
                         "for a rank count that was not asked for" % (args.gpus, env_world))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    under_launcher = "TORCHELASTIC_RUN_ID" in os.environ or "LOCAL_RANK" in os.environ
    if "SCILMM_HOST_THREADS" not in os.environ and (under_launcher or "OMP_NUM_THREADS" not in os.environ):
        # Host thread teams of the library (analysis, value permutation, device plan).  torch.distributed.run exports
        # OMP_NUM_THREADS=1 to every worker unless the caller set it: a generic default of the launcher, not a choice about this
        # job -- honoured, it would run the 20 s analysis of the 1M cohort on ONE thread (minutes).  So under a launcher the team
        # size is always set here, from the CPUs this job may really use (affinity mask and cgroup quota): rank 0, which simulates
        # and analyses the cohort while the others wait at a barrier, gets all of them, the others their share (they build their
        # device plans at the same time).  A user's SCILMM_HOST_THREADS wins; without a launcher so does OMP_NUM_THREADS.
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        cpus = _effective_cpus()
        os.environ["SCILMM_HOST_THREADS"] = str(cpus if rank == 0 else max(1, cpus // max(1, local_world)))
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)  # one rank per GPU under the driver; the modulo only matters in rehearsals
    dev = torch.device("cuda", local_rank % ndev)
    # "nccl" is RCCL.  SCILMM_BENCH_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks
    backend = os.environ.get("SCILMM_BENCH_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    rdev = dev if backend == "nccl" else torch.device("cpu")  # where the reduced scalars live

    def agree_min(x):
        """the same (minimum) integer on every rank"""
        if world == 1:
            return int(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0])

    from scilmm_amd.factor import Symbolic
    want_cpu = world == 1 and rank == 0 and not args.no_cpu_baseline
    t0 = time.time()
    # One cohort for the whole job.  Rank 0 simulates it and hands it to the other ranks through /dev/shm (N
    # simultaneous simulations of a 1M pedigree would need N x 40 GB of host memory and N x the cores).
    seed = int(os.environ.get("SCILMM_BENCH_SEED", "0"))
    comps = None  # extra variance-component matrices on the host (K = 3: the dominance matrix; N > 1: on rank 0 only)
    par = None    # N > 1: the cohort's parent table -- every rank builds the values of A (and D) in its own HBM from it
    if world == 1:
        A, C, y = build_problem(args.workload, seed=seed, components=args.components)
        if args.components != "A":
            comps = A[1:]
            A = A[0]
    else:
        # (hand-off directory: memory-backed /dev/shm when it has the room -- 26 GB at the 1M config -- else the temp directory)
        # (the pattern of A + the parent table -- no values -- and the image of the analysis the ranks share: 24 bytes per entry of
        #  tril(A) for its assembly maps alone)
        need = {"3m": 130e9, "1m": 45e9, "300k": 12e9}.get(args.workload, 3e9)
        import shutil
        import tempfile
        roots = [d for d in ("/dev/shm", os.environ.get("TMPDIR") or tempfile.gettempdir(), os.getcwd())
                 if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need]
        if not roots:
            raise SystemExit("bench.py: no directory with %.0f GB free for the rank hand-off files" % (need / 1e9))
        roots = roots[:1]
        dist.broadcast_object_list(roots, src=0)  # (rank 0's choice: free space is a moving number)
        shm = "%s/scilmm_bench_%s_%s" % (roots[0], os.environ.get("MASTER_PORT", "0"), args.workload)
        from scilmm_amd.factor import PatternCSR
        if rank == 0:
            # (rank 0 keeps the host matrices: it checks the residual of the last solve against them -- and with it the values
            #  the ranks built on their devices)
            A, C, y, par = build_problem(args.workload, seed=seed, components=args.components, parents=True)
            if args.components != "A":
                comps = A[1:]
                A = A[0]
            np.save(shm + "_indptr.npy", A.indptr); np.save(shm + "_indices.npy", A.indices)
            np.save(shm + "_par.npy", par); np.save(shm + "_C.npy", C); np.save(shm + "_y.npy", y)
            Apat = PatternCSR(A.indptr, A.indices, A.shape[0])
        dist.barrier()
        if rank != 0:
            C, y, par = np.load(shm + "_C.npy"), np.load(shm + "_y.npy"), np.load(shm + "_par.npy")
            A = None
            Apat = PatternCSR(np.load(shm + "_indptr.npy"), np.load(shm + "_indices.npy"), y.size)
        dist.barrier()
        if rank == 0:
            for suffix in ("indptr", "indices", "par", "C", "y"):
                os.remove(shm + "_%s.npy" % suffix)
    n = y.size
    t_gen = time.time() - t0
    t0 = time.time()
    eng = None
    if world == 1:
        # (SCILMM_BENCH_SYM="relax_w1=32,relax_z2=0.2": analysis options for tuning experiments; default = library defaults)
        sym_opts = {}
        for kv in filter(None, os.environ.get("SCILMM_BENCH_SYM", "").split(",")):
            k, v = kv.split("=")
            sym_opts[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
        sym = Symbolic([A] + (comps or []) + [sp.identity(n, format="csr")], **sym_opts)
    else:
        from scilmm_amd.dist import HipChainEngine
        # one analysis per NODE: rank 0 analyses the pattern and leaves its image in /dev/shm, the others load it
        cache = "%s/scilmm_bench_sym_%s" % (roots[0], os.environ.get("MASTER_PORT", "0"))
        # value-less patterns: A (and D on A's pattern, as the reference builds it) + the identity
        mats_e = [Apat] + ([Apat] if args.components != "A" else []) + [sp.identity(n, format="csr")]
        # (with a cache HipChainEngine lets rank 0 analyse and publish the image before the others load it)
        eng = HipChainEngine(mats_e, rank, world, dist, dev, cache=cache)
        sym = eng.sym
        # the values of A by the tabular recursion, those of D from A's resident slots: nothing but the pattern and the parent
        # table reached this rank (scilmm_ibd_values_device, scilmm_dominance_values_device)
        sym.ibd_values_from_pedigree(0, par)
        if args.components != "A":
            sym.dominance_values_from(1, 0, par)
        dist.barrier()
        if rank == 0:
            shutil.rmtree(cache, ignore_errors=True)
        # N processes of one node each hold the cohort and the analysis: drop what an evaluating rank no longer needs
        sym.release_host_maps()
        del Apat, mats_e
    t_sym = time.time() - t0
    info = sym.info()
    # The timed steps run WITHOUT the engine's HIP-event brackets (14 timed events per level: 4 - 5 ms of a 52 ms factorization at
    # the 100k config, tools/enqueue_time.py); the per-class figures of the line come from ONE extra, untimed evaluation with the
    # brackets on, the clean per-launch figure from one more with the look-ahead launches serialised.  --serialised: every
    # evaluation in mode 2 (a measurement mode); --no-engine-profiling: no bracketed evaluation at all.
    prof_mode = 0 if args.no_engine_profiling else (2 if args.serialised else 1)
    timed_mode = 2 if args.serialised else 0
    if timed_mode:
        sym.set_profiling(timed_mode)
    if args.no_engine_profiling or args.serialised:
        args.no_clean_profile = True
    if args.front_bits == 32:
        sym.set_front_precision(32)

    c, s = C.shape[1], 100
    r = c + 1 + s
    rng = np.random.default_rng(100)  # the same right-hand sides on every rank
    B_host = np.hstack([C, y[:, None], rng.standard_normal((n, s))])
    dB = torch.from_numpy(B_host).to(dev)   # every rank takes part in every column of the (collective) sweep
    dX = torch.empty_like(dB)
    if args.front_bits == 32 and world == 1:
        dY, dRes = torch.empty_like(dB), torch.empty_like(dB)
    torch.cuda.synchronize()

    state = {"fac": None}
    logdets = []

    k3 = args.components != "A"
    refine_steps = 2 if args.front_bits == 32 else 0
    ev = None
    if eng is not None and refine_steps:
        from scilmm_amd.dist import DistributedEvaluator
        ev = DistributedEvaluator(eng, [None] * (3 if k3 else 2), C, y, rank, world, dist, refine_steps=refine_steps)
    refine_wall = [0.0]

    def sigma2_of(i):
        if k3:  # K = 3: sigma2 = (additive, dominance, residual) around SURVEY's (0.3, 0.1, 0.6)
            return [0.3 + 0.01 * (i % 3), 0.1, 0.6 - 0.01 * (i % 3)]
        return [0.4 + 0.01 * (i % 3), 0.6 - 0.01 * (i % 3)]

    def step(i):
        if eng is not None:
            eng.factorize(sigma2_of(i))  # the first evaluation also builds the device plan
            state["fac"] = eng.fac
        elif state["fac"] is None:
            state["fac"] = sym.factorize(sigma2_of(i))
        else:
            state["fac"].refactorize(sigma2_of(i))
        fac = state["fac"]
        logdets.append(fac.logdet())
        vp = ctypes.c_void_p
        if ev is not None:
            # fp32-product fronts on the multi-rank path: the collective sweep + the refinement sweeps against the exact V
            # (column-split SpMMs + one all-reduce, one more collective sweep each: scilmm_amd/dist.py)
            t_r = time.perf_counter()
            dX.copy_(ev._refined_solve(np.asarray(sigma2_of(i)), dB, True))
            torch.cuda.synchronize()
            refine_wall[0] += time.perf_counter() - t_r
            return
        fac.solve_dev(vp(dB.data_ptr()), r, vp(dX.data_ptr()))
        sym.sync()  # (N > 1: every rank ends with all 103 solution columns, as the REML evaluation needs)
        if refine_steps:
            # fp32-product fronts: the step delivers the solve the REML evaluation uses -- refined on the device against the
            # exact V = sum_k sigma2_k A_k (what scilmm_amd.SparseCholesky._finish_on_device does): K SpMMs + one more sweep each
            t_r = time.perf_counter()
            s2 = sigma2_of(i)
            for _ in range(refine_steps):
                dRes.copy_(dB)
                for k in range(len(s2)):
                    torch.cuda.synchronize()
                    sym.spmm_dev(k, vp(dX.data_ptr()), r, vp(dY.data_ptr()))
                    sym.sync()
                    dRes.sub_(dY, alpha=float(s2[k]))
                torch.cuda.synchronize()
                fac.solve_dev(vp(dRes.data_ptr()), r, vp(dY.data_ptr()))
                sym.sync()
                dX.add_(dY)
            torch.cuda.synchronize()
            refine_wall[0] += time.perf_counter() - t_r

    if args.dump_maps and rank == 0:
        with open("/proc/self/maps") as fi, open(args.dump_maps, "w") as fo:
            fo.write(fi.read())
    t0 = time.time()
    step(0)  # the plan-building evaluation always runs before the timed region: it is the first warm-up step
    t_first = time.time() - t0
    warm_done = 1
    tm = sym.timing()  # HIP-event time of that evaluation's kernels = a good estimate of a steady-state step
    t_step_est = (tm["assemble_ms"] + tm["factor_ms"] + tm["solve_fwd_ms"] + tm["solve_bwd_ms"]) / 1e3 * 1.05 + 0.01
    reserve = CPU_BASELINE_RESERVE_S if want_cpu else 0.0
    if world == 1 and not args.no_clean_profile:
        reserve += 1.25 * t_step_est  # the serialised profiling evaluation after the timed region
    if prof_mode == 1:
        reserve += 1.1 * t_step_est   # the bracketed evaluation (per-class figures) after the timed region
    remaining = args.budget_s - (time.time() - T_PROCESS_START) - reserve - 10.0
    afford = int(remaining / max(t_step_est, 1e-9))
    steps = max(1, min(args.steps, afford))
    extra_warm = max(0, min(args.warmup - warm_done, afford - steps))
    steps = agree_min(steps)
    extra_warm = agree_min(extra_warm)
    for i in range(extra_warm):
        step(1 + i)
    warm_done += extra_warm
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prof = {"dense_ms": 0.0, "n_dense_launches": 0, "update_union_ms": 0.0, "update_ms": 0.0, "potrf_ms": 0.0, "trsm_ms": 0.0, "reduce_cells_ms": 0.0, "assemble_ms": 0.0, "factor_ms": 0.0,
            "solve_fwd_ms": 0.0, "solve_bwd_ms": 0.0, "n_update_launches": 0, "n_launches": 0}
    refine_wall[0] = 0.0
    for i in range(steps):
        step(i)
        t = sym.timing()
        for k in prof:
            prof[k] += t[k]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_used = float(total_b - free_b)   # this rank's device: factor share, plan, values, work buffers, allocator caches
    if world > 1:
        hmax = torch.tensor([hbm_used], dtype=torch.float64, device=rdev)
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        hbm_used = float(hmax[0])
    nnzL_total = float(info.nnzL)  # ONE cohort, whatever the number of ranks
    logdet_total = logdets[-1]
    bracketed_factor_ms = None
    if prof_mode == 1:
        # ONE more, untimed evaluation under the default schedule with the kernel classes bracketed by HIP events: the per-class
        # sums and the overlapped per-launch figures of the line (scaled to the K timed steps, which ran without the brackets)
        sym.set_profiling(1)
        step(steps - 1)
        cls = sym.timing()
        sym.set_profiling(0)
        logdets.pop()
        bracketed_factor_ms = cls["factor_ms"]
        for k in ("dense_ms", "n_dense_launches", "update_union_ms", "update_ms", "potrf_ms", "trsm_ms", "reduce_cells_ms", "n_update_launches"):
            prof[k] = cls[k] * steps

    # ONE more, untimed evaluation with every look-ahead launch on a single side stream: consecutive launches of the
    # dominant kernel then do not overlap each other, which gives the clean per-launch duration `roofline.achieved` asks
    # for (in the timed steps two launches share the chip at any time, so a launch lasts about twice what it needs)
    clean = None
    if world == 1 and not args.no_clean_profile:
        sym.set_profiling(2)
        step(steps - 1)  # (same sigma2 as the last timed step: the residual check below is of this solve)
        clean = sym.timing()
        sym.set_profiling(timed_mode)
        logdets.pop()
    # residual check of the last solve (outside the timed region)
    fac = state["fac"]
    X = dX[:, :r].cpu().numpy()
    s2 = sigma2_of(steps - 1)
    probe = [0, c, r - 1]  # first covariate column, the phenotype, the last simulated vector (last rank's share)
    resid = 0.0
    if rank == 0:
        VX = s2[0] * (A @ X[:, probe]) + s2[-1] * X[:, probe]
        for kq, Mq in enumerate(comps or []):
            VX += s2[1 + kq] * (Mq @ X[:, probe])
        resid = float(np.abs(VX - B_host[:, probe]).max() / np.abs(B_host[:, probe]).max())

    if rank == 0:
        K = steps
        # (a --pmc pass of the 1M workload does not finish inside the pool's per-call limit -- counter collection
        # serialises the 8000 launches of a factorization; the 300k pass is on file: profiles/r2_traffic_300k.json)
        traffic, traffic_src = None, None
        tpath = next((q for q in (os.path.join(ROOT, "profiles", "r%d_traffic_%s.json" % (rr, args.workload)) for rr in (4, 3, 2))
                      if os.path.exists(q)), "")
        if tpath:
            # HBM bytes of the update kernel from a separate rocprofv3 --pmc pass of this workload (a PMC pass cannot
            # run inside the timed bench); per launch like `achieved`; the file names its command and corrections
            traffic = json.load(open(tpath))["bytes_per_launch"]
            traffic_src = "%s (offline rocprofv3 --pmc passes of this workload, not measured by this run)" % os.path.relpath(tpath, ROOT)
        iso = os.path.join(ROOT, "profiles", "r3_traffic_%s_isolated_kernel.json" % args.workload)
        traffic_per_flop = None
        if traffic is None and os.path.exists(iso):
            # no in-situ PMC pass exists for this workload (rocprofv3 --pmc aborts at 1M: the file says how): HBM bytes per
            # flop of the dominant kernel from a PMC pass of that kernel ALONE on items of this workload's shape, scaled to
            # this run's flops per launch below
            traffic_per_flop = 1.0 / json.load(open(iso))["flop_per_byte"]
            traffic_src = ("%s: bytes per flop of the dominant kernel alone at this workload's shape (rocprofv3 --pmc FETCH_SIZE / "
                           "WRITE_SIZE passes of csrc/tools/dense_bench2) x this run's flops per launch; an in-situ --pmc pass "
                           "of this workload aborts inside the profiler" % os.path.relpath(iso, ROOT))
        upd_s = prof["update_ms"] / 1e3
        n_upd = max(prof["n_update_launches"], 1)
        ach_all = info.update_flops * K / world / max(upd_s, 1e-12) / 1e12
        dense_on = prof["n_dense_launches"] > 0
        if dense_on:
            # dominant kernel = k_dense: ITS algorithmic flops (tail x tail updates, true structure) over ITS launches
            kern = (("k_dense32 (fp32 products, fp64 sums: update of the dense tail by the dense tail, fp64 operands)"
                     if os.environ.get("SCILMM_SHADOW") == "0" else
                     "k_dense_h (fp32 products out of an fp32 shadow of the tail panels, fp64 sums every 256 k: update of the dense "
                     "tail by the dense tail)") if args.front_bits == 32 else
                    "k_dense_b (fp64 MFMA update of the dense tail by the dense tail: A fragments from registers, B by LDS-DMA, "
                    "both streams software-pipelined in the wave)")
            # (N > 1: THIS rank's launches -- batches + late items -- against its 1 / N share of the tail's flops: the panels
            #  are owned block-cyclically, so the shares are equal to a fraction of a percent)
            flops_k, n_k, ms_k = info.dense_flops * K / world, prof["n_dense_launches"], prof["dense_ms"]
        else:
            kern = "k_update2<true> (fp64 MFMA supernodal update)"
            flops_k, n_k, ms_k = info.update_flops * K, n_upd, prof["update_ms"]
        ach_overlapped = flops_k / max(ms_k / 1e3, 1e-12) / 1e12
        ach = ach_overlapped
        clean_note = ("launch durations of one untimed evaluation under the default schedule (two launches overlap at any time)" if not args.serialised else
                      "the timed steps themselves, run with the look-ahead launches serialised on one stream (--serialised): launch "
                      "durations do not overlap each other; rocprofv3 --kernel-trace --stats of this command shows the same average")
        if clean is not None:
            # the same kernel class in the serialised evaluation: its algorithmic flops over ITS summed launch durations
            c_ms = clean["dense_ms"] if dense_on else clean["update_ms"]
            c_n = clean["n_dense_launches"] if dense_on else clean["n_update_launches"]
            if c_ms > 0 and c_n > 0:
                ach = (info.dense_flops if dense_on else info.update_flops) / (c_ms / 1e3) / 1e12
                n_k, ms_k, flops_k = c_n, c_ms, (info.dense_flops if dense_on else info.update_flops)
                clean_note = ("one untimed evaluation after the timed region with the look-ahead launches serialised on one "
                              "stream (factorization %.3f s instead of %.3f s): launch durations do not overlap each other"
                              % (clean["factor_ms"] / 1e3, prof["factor_ms"] / K / 1e3))
        solve_s = (prof["solve_fwd_ms"] + prof["solve_bwd_ms"]) / 1e3 / K
        solve_bytes = 16.0 * info.nnzL + 32.0 * n * r
        solve_flops = 4.0 * info.nnzL * r
        reasons = []
        if steps < args.steps or warm_done != args.warmup:
            reasons.append("step counts budgeted by wall clock: %.1f s per step, %.0f s budget for the whole process "
                           "(generation %.0f s, analysis %.0f s, first evaluation incl. device plan %.0f s%s)"
                           % (t_step_est, args.budget_s, t_gen, t_sym, t_first,
                              ", %.0f s reserved for the CPU baseline" % reserve if reserve else ""))
        if traffic is None and traffic_per_flop is not None:
            traffic = traffic_per_flop * flops_k / max(n_k, 1)
        out = {
            "metric": ("REML factorize+solve nnz(L)/s (simulated pedigree, fp64)" if not k3 else
                       "REML factorize+solve nnz(L)/s (simulated pedigree, K=3 A+D+I, fp64: configs[4]'s model, NOT the headline)") if args.front_bits == 64 else
                      "REML factorize+solve nnz(L)/s (simulated pedigree, fp64 factor with fp32 MFMA fronts: configs[4] arithmetic, NOT the fp64 headline)",
            "value": nnzL_total * K / elapsed,
            "unit": "nnz(L)/s",
            "n_gpus": world, "steps": K, "warmup": warm_done,
            "steps_requested": args.steps, "warmup_requested": args.warmup,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.front_bits == 64 else "f64 sums / f32 MFMA products in the dense tail", "data": "synthetic",
            "config": {"workload": "simulated pedigree %s (n=%d after unrelated-drop, sparsity_factor %g), %s, "
                                   "r=%d fused right-hand sides" % (args.workload, n, WORKLOADS[args.workload][1],
                                                                    "K=3 (A + D + I; D built on the device)" if k3 else "K=2 (A + I)", r),
                       "baseline_config": ("configs[4]" if (args.workload == "3m" and k3 and args.front_bits == 32 and world == 8) else
                                           "configs[3]" if (args.workload == "1m" and not k3 and args.front_bits == 64 and world == 8) else
                                           {"10k": "configs[0]", "100k": "configs[1]", "1m": "configs[2]"}.get(args.workload, "probe")
                                           if (world == 1 and not k3 and args.front_bits == 64) else "probe"),
                       "step_budget": "; ".join(reasons) if reasons else "requested counts run unchanged",
                       "n": n, "nnz_tril_A": int((A.nnz + n) // 2), "nnzL": int(info.nnzL),
                       "nnzL_stored": int(info.nnzL_stored), "factor_flops": info.flops, "nsuper": info.nsuper,
                       "nlevels": info.nlevels,
                       "parallelism": "1 GPU" if world == 1 else "one cohort over %d ranks: dense-tail panels 1-D block-cyclic, fan-out "
                                      "with rank-local storage (panel broadcast over RCCL into a ring, batched group updates), "
                                      "prelude replicated, collective sweeps" % world,
                       "local_factor_GB": (8e-9 * eng.local_factor_doubles) if eng is not None else 8e-9 * info.nnzL_stored,
                       "hbm_in_use_GB_max_over_ranks": hbm_used / 1e9,
                       "values": "A by scilmm_ibd_values_device%s in every rank's HBM from the pattern + parent table" % (
                           ", D by scilmm_dominance_values_device" if k3 else "") if eng is not None else "uploaded from the host",
                       "refinement": ("%d sweeps per step against the exact V on the device (fp32-product fronts), %.3f s per step, inside "
                                      "the timed step" % (refine_steps, refine_wall[0] / K)) if refine_steps else "none (fp64 factor)",
                       "seconds_per_step": elapsed / K,
                       "factorize_ms": prof["factor_ms"] / K, "assemble_ms": prof["assemble_ms"] / K,
                       "solve_ms": (prof["solve_fwd_ms"] + prof["solve_bwd_ms"]) / K,
                       "solve_fwd_ms": prof["solve_fwd_ms"] / K, "solve_bwd_ms": prof["solve_bwd_ms"] / K,
                       "factorize_tflops": info.flops * K / max(prof["factor_ms"] / 1e3, 1e-12) / 1e12,
                       "solve_hbm_gbs": solve_bytes / max(solve_s, 1e-12) / 1e9,
                       "solve_hbm_frac": solve_bytes / max(solve_s, 1e-12) / 1e9 / HBM_PEAK_GBS,
                       "solve_tflops": solve_flops / max(solve_s, 1e-12) / 1e12,
                       "solve_mfma_frac": solve_flops / max(solve_s, 1e-12) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                       "update_ms": prof["update_ms"] / K, "potrf_ms": prof["potrf_ms"] / K,
                       "trsm_ms": prof["trsm_ms"] / K, "main_stream_reduce_cells_ms": prof["reduce_cells_ms"] / K,
                       "launches_per_factorize": prof["n_launches"] / K,
                       "timed_steps_instrumentation": ("none: the timed steps run without the engine's HIP-event brackets; per-class figures from "
                                                       "one untimed bracketed evaluation (factorization %.3f ms with the brackets on)" % bracketed_factor_ms)
                                                      if bracketed_factor_ms is not None else
                                                      ("mode 2 brackets in every step (--serialised)" if args.serialised else "none"),
                       "symbolic_s": t_sym, "generate_s": t_gen, "first_evaluation_s": t_first,
                       "logdet": logdet_total, "solve_residual": resid},
            "roofline": {"bound": "mfma", "kernel": kern,
                         # (fp32-product fronts: the dominant kernel's products run on the fp32 matrix pipe, 157.3 TFLOP/s)
                         "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS if (args.front_bits == 32 and dense_on) else FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s",
                         "frac": ach / (FP32_MFMA_PEAK_TFLOPS if (args.front_bits == 32 and dense_on) else FP64_MFMA_PEAK_TFLOPS),
                         "traffic": traffic, "traffic_source": traffic_src,
                         # launches of consecutive levels overlap on two streams: the same flops over the time during
                         # which at least one update launch was running (not the figure the contract asks for)
                         "achieved_over_busy_time": (info.update_flops * K / max(prof["update_union_ms"] / 1e3, 1e-12) / 1e12) if world == 1 else None,
                         "rank": "rank 0 of %d: its own launches against 1/%d of the flops" % (world, world) if world > 1 else "the one GPU",
                         "flops_per_launch": flops_k / max(n_k, 1),
                         "avg_launch_ms": ms_k / max(n_k, 1),
                         "launches": int(n_k),
                         "measured_on": clean_note,
                         "achieved_with_overlapping_launches": ach_overlapped,
                         "note": "under the default schedule launches of consecutive levels overlap pairwise on two streams, so a launch's "
                                 "duration is about twice what it needs alone (achieved_with_overlapping_launches); "
                                 "achieved_over_busy_time counts that overlapped time once; `achieved` is the per-launch figure "
                                 "without overlap",
                         "all_update_kernels": {"achieved": ach_all, "launches": int(n_upd), "summed_launch_ms": prof["update_ms"]}},
        }
        if want_cpu:
            try:
                del dB, dX
                state["fac"] = None
                del fac
                if args.workload == CPU_BASELINE_WORKLOAD:
                    Ac = A
                else:
                    del sym
                    Ac, _, _ = build_problem(CPU_BASELINE_WORKLOAD, seed=0)
                out["cpu_baseline"] = cpu_baseline(Ac, r, CPU_BASELINE_WORKLOAD)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "nnz(L)/s", "cores": len(os.sched_getaffinity(0)),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        else:
            out["cpu_baseline"] = None
        if args.no_engine_profiling:
            out["roofline"] = None   # (no HIP-event brackets were taken: nothing to quote)
        if args.serialised:
            out["config"]["mode"] = "--serialised: a measurement mode (look-ahead launches on one stream), slower than the default schedule"
        out["config"]["process_wall_s"] = time.time() - T_PROCESS_START
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
