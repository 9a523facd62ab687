"""N > 1 paths on the CPU (gloo): the component-sharded REML evaluation (world 2) and the distributed dense tail -- fan-out
with rank-local storage -- at world 2, 4 and 8 must reproduce the single-process oracle evaluation (same global
permutation, same np.random stream) to rounding."""
import os
import socket

import numpy as np
import scipy.sparse as sp

from tests.helpers import small_pedigree, small_pedigree_k3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleEngine(object):
    def __init__(self, mats, perm):
        self.mats, self.perm = mats, perm

    def factorize(self, s2):
        from oracle import oracle as O
        V = sum(a * m for a, m in zip(s2, self.mats)).tocsr()
        return O.OracleFactor(V, self.perm)

    def quadforms(self, k, Q):
        from oracle import oracle as O
        return O.quadforms(self.mats[k], Q)


def _problem():
    A, sex = small_pedigree(3000, 0.003, 5)
    n = A.shape[0]
    rng = np.random.default_rng(1)
    y = rng.standard_normal(n)
    C = np.stack([(sex - sex.mean()) / sex.std(), np.ones(n)], axis=1)
    return [A, sp.eye(n).tocsr()], C, y


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scilmm_amd.factor import Symbolic
    from scilmm_amd.shard import ShardedEvaluator
    mats, C, y = _problem()
    perm = Symbolic(mats, upload=False).P()
    ev = ShardedEvaluator(mats, C, y, perm, rank, world, _OracleEngine, dist=dist)
    res = []
    for reml in (True, False):
        np.random.seed(4)
        res.append(ev.evaluate(np.log([0.45, 0.5]), reml=reml, sim_num=50))
    if rank == 0:
        np.savez(out, nll=np.array([r[0] for r in res]), grad=np.array([r[1] for r in res]), load=ev.load,
                 nloc=ev.y.size)
    dist.destroy_process_group()


def test_two_rank_sharded_evaluation_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle import reml_oracle as RO
    from scilmm_amd.factor import Symbolic
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    mats, C, y = _problem()
    perm = Symbolic(mats, upload=False).P()
    for i, reml in enumerate((True, False)):
        np.random.seed(4)
        nll, grad = RO.evaluate(np.log([0.45, 0.5]), mats, C, y, reml, 50, perm=perm)
        assert abs(got["nll"][i] - nll) < 1e-10 * abs(nll)
        assert np.abs(got["grad"][i] - grad).max() < 1e-8 * np.abs(grad).max()
    # both ranks got work and the split is not degenerate
    assert 0 < int(got["nloc"]) < y.size
    assert got["load"].min() > 0


def test_partition_keeps_components_whole():
    from scipy.sparse.csgraph import connected_components
    from scilmm_amd.shard import component_partition
    mats, _, _ = _problem()
    owner, load = component_partition(mats, 4)
    _, label = connected_components(mats[0], directed=False)
    for c in np.unique(label):
        assert np.unique(owner[label == c]).size == 1
    assert load.shape == (4,)


# ---- level 2: ONE connected component split across the ranks: distributed dense tail, fan-out with rank-local storage
#      (scilmm_amd/dist.py; the CPU stand-in oracle/dist_cpu.py reads the distribution rule from the library)

def _tail_problem(which):
    n0, sf, seed = {"24panels": (20000, 0.01, 1), "39panels": (30000, 0.01, 2), "24panels_k3": (20000, 0.01, 1)}[which]
    if which.endswith("_k3"):
        A, D, sex = small_pedigree_k3(n0, sf, seed)   # K = 3: additive + dominance + identity (BASELINE configs[4]'s model)
        mats = [A, D]
    else:
        A, sex = small_pedigree(n0, sf, seed)
        mats = [A]
    n = A.shape[0]
    rng = np.random.default_rng(2)
    y = rng.standard_normal(n)
    C = np.stack([(sex - sex.mean()) / sex.std(), np.ones(n)], axis=1)
    return mats + [sp.eye(n).tocsr()], C, y


def _sigma2_of(which):
    return [0.3, 0.15, 0.5] if which.endswith("_k3") else [0.45, 0.5]


def _tail_worker(rank, world, port, out, which, group, refine=None):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    from threadpoolctl import threadpool_limits
    threadpool_limits(limits=1)  # `world` processes share this machine's cores: one BLAS thread each
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SCILMM_HOST_THREADS"] = "1"
    if group:
        os.environ["SCILMM_TUNING"] = "1"
        os.environ["SCILMM_DIST_GROUP"] = str(group)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.dist_cpu import CpuChainEngine
    from scilmm_amd.dist import DistributedEvaluator
    mats, C, y = _tail_problem(which)
    eng = CpuChainEngine(mats, rank, world, dist)
    ev = DistributedEvaluator(eng, mats, C, y, rank, world, dist, refine_steps=refine)
    res = []
    for reml in (True, False):
        np.random.seed(4)
        res.append(ev.evaluate(np.log(_sigma2_of(which)), reml=reml, sim_num=20))
    np.savez(out % rank, nll=np.array([r[0] for r in res]), grad=np.array([r[1] for r in res]), first=eng.first, Wg=eng.Wg,
             refinement=np.array(ev.last_refinement),
             G=eng.G, nT=eng.nT, computed=eng.panels_computed, received=eng.panels_received, batches=eng.batches,
             local=eng.local_doubles, total=eng.global_doubles, ns=eng.ns, perm=eng.perm, owner=eng.owner,
             collectives=eng.collectives)
    dist.destroy_process_group()


def _check_tail_run(tmp_path, world, which, group, reference, refine=None):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from oracle import reml_oracle as RO
    from scilmm_amd.factor import Symbolic
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_tail_worker, args=(world, _free_port(), out, which, group, refine), nprocs=world, join=True)
    got = [np.load(out % r) for r in range(world)]
    mats, C, y = _tail_problem(which)
    perm = got[0]["perm"]
    if reference == "port":
        sym = Symbolic(mats, upload=False)
        arrays, colptr = sym.arrays(), sym.get("pat_colptr")
        assert np.array_equal(arrays["perm"], perm)
        factor_of = lambda V: O.CPUPortFactor(arrays, colptr, V)
    else:
        factor_of = None  # the simplicial C oracle
    for i, reml in enumerate((True, False)):
        np.random.seed(4)
        nll, grad = RO.evaluate(np.log(_sigma2_of(which)), mats, C, y, reml, 20, perm=perm, factor_of=factor_of)
        for g in got:  # every rank ends with the same numbers
            assert abs(g["nll"][i] - nll) < 1e-10 * abs(nll)
            assert np.abs(g["grad"][i] - grad).max() < 1e-8 * np.abs(grad).max()
    # the split is real
    nT, Wg, G, first, ns = (int(got[0][k]) for k in ("nT", "Wg", "G", "first", "ns"))
    assert Wg % world == 0 and G == 4 * Wg and nT > 2 * Wg  # at least one batch reaches a target
    sn_loff = Symbolic(mats, upload=False).get("sn_loff")
    sizes = np.diff(np.append(sn_loff, int(got[0]["total"])))[:ns] if sn_loff.size == ns else np.diff(sn_loff)
    slot = int(((sizes[first:] + 1) // 2 * 2).max())
    for r, g in enumerate(got):
        mine = [j for j in range(nT) if j % world == r]
        assert int(g["computed"]) == 2 * len(mine)              # two evaluations; each tail panel factored by its owner only
        assert int(g["received"]) == 2 * (nT - len(mine))       # ... and received by everybody else
        assert int(g["batches"]) == 2 * ((nT + Wg - 1) // Wg)
        assert np.array_equal(g["owner"][first:], np.arange(nT) % world) and np.all(g["owner"][:first] == -1)
        # rank-local storage = prelude + own tail panels + ring (NOT the whole factor)
        own = int(((sizes[first:][mine] + 1) // 2 * 2).sum())
        assert int(g["local"]) == int(sn_loff[first]) + own + min(G, nT) * slot
    # (at these toy sizes the ring -- min(G, nT) slots of the LARGEST panel -- outweighs the saving; at 1M, 1393 panels
    #  of which 32 ring slots: 123 GB -> 15.4 + 4.8 GB per rank at world 8)
    return got


def test_two_rank_distributed_tail_with_ring_reuse(tmp_path):
    """24 tail panels over 2 ranks in groups of 2 (ring of 8 slots, re-used three times; 12 batches): nll and gradient of a
    REML / ML evaluation match the single-process simplicial oracle (same P, same RNG stream)."""
    _check_tail_run(tmp_path, 2, "24panels", 2, "simplicial")


def test_four_rank_distributed_tail(tmp_path):
    """39 tail panels over 4 ranks, default rule (groups of 8, ring of 32 slots: re-used)."""
    _check_tail_run(tmp_path, 4, "39panels", 0, "port")


def test_eight_rank_distributed_tail(tmp_path):
    """39 tail panels over 8 ranks (BASELINE configs[3]'s rank count), default rule."""
    got = _check_tail_run(tmp_path, 8, "39panels", 0, "port")
    # every rank issued the same sequence of collectives
    assert len({int(g["collectives"]) for g in got}) == 1


def test_eight_rank_k3_with_refinement_sweeps(tmp_path):
    """BASELINE configs[4]'s model and control flow at its rank count: K = 3 (A + D + I), 24 tail panels over 8 ranks, and
    the refinement sweeps a factor with fp32-product fronts needs (residual V x by K column-split SpMMs + one all-reduce,
    one more collective solve per sweep) -- forced here on the fp64 CPU factor, where a sweep must change nothing: the
    evaluation still matches the single-process oracle, and the corrections are at rounding level."""
    got = _check_tail_run(tmp_path, 8, "24panels_k3", 0, "port", refine=2)
    assert len({int(g["collectives"]) for g in got}) == 1
    for g in got:
        assert g["refinement"].shape == (2,) and g["refinement"].max() < 1e-12


def _npd_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SCILMM_TUNING"] = "1"
    os.environ["SCILMM_DIST_GROUP"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from oracle.dist_cpu import CpuChainEngine
    rng = np.random.default_rng(0)
    n = 700
    Gm = rng.standard_normal((n, n))
    A = sp.csr_matrix(Gm @ Gm.T / n)       # dense: every front is a tail panel (6 of them)
    I = sp.eye(n).tocsr()
    eng = CpuChainEngine([A, I], rank, world, dist, perm=np.arange(n))
    raised = -2
    try:
        eng.factorize([1.0, -0.5])           # indefinite: the pivot fails inside ONE rank's panel
    except O.NotPositiveDefinite as e:
        raised = int(e.args[0])
    # ... and the engine is usable afterwards, on every rank, on a definite matrix (all-tail: dist from level 0)
    eng.factorize([1.0, 0.5])
    b = np.random.default_rng(1).standard_normal((n, 3))
    x = eng.solve(b)
    V = (A + 0.5 * I).toarray()
    np.savez(out % rank, raised=raised, resid=np.abs(V @ x - b).max(), logdet=eng.logdet(), ref=np.linalg.slogdet(V)[1],
             first=eng.first, nT=eng.nT)
    dist.destroy_process_group()


def test_not_positive_definite_is_raised_on_every_rank(tmp_path):
    """only the owner of the failing panel saw the bad pivot; now the status is agreed on by all ranks
    (all-reduce MIN), so every rank raises and none is left waiting in the next collective.  Also covers a matrix whose
    EVERY front is distributed (dense: the distributed part starts at level 0)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "npd%d.npz")
    mp.spawn(_npd_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = [np.load(out % r) for r in range(2)]
    assert got[0]["raised"] >= 0 and got[0]["raised"] == got[1]["raised"]
    for g in got:
        assert int(g["first"]) == 0 and int(g["nT"]) >= 4
        assert float(g["resid"]) < 1e-9 and abs(float(g["logdet"]) - float(g["ref"])) < 1e-9 * abs(float(g["ref"]))


def test_column_chunks_cover_all_columns():
    from scilmm_amd.dist import column_chunks
    for r in (1, 5, 103, 104):
        for world in (1, 2, 3, 8):
            ch = column_chunks(r, world)
            assert ch[0][0] == 0 and ch[-1][1] == r and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
            assert max(c1 - c0 for c0, c1 in ch) - min(c1 - c0 for c0, c1 in ch) <= 1
