"""N > 1 path on the CPU: world_size-2 gloo run of the component-sharded REML evaluation must reproduce the
single-process oracle evaluation (same global permutation, same np.random stream) to rounding."""
import os
import socket

import numpy as np
import scipy.sparse as sp

from tests.helpers import small_pedigree


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


# ---- level 2: ONE connected component split across the ranks (distributed separator chain, panel broadcasts)

def _chain_problem():
    A, sex = small_pedigree(6000, 0.01, 3)
    n = A.shape[0]
    rng = np.random.default_rng(2)
    y = rng.standard_normal(n)
    C = np.stack([(sex - sex.mean()) / sex.std(), np.ones(n)], axis=1)
    return [A, sp.eye(n).tocsr()], C, y


def _chain_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.dist_cpu import CpuChainEngine
    from scilmm_amd.dist import DistributedEvaluator
    mats, C, y = _chain_problem()
    eng = CpuChainEngine(mats, rank, world, dist)
    ev = DistributedEvaluator(eng, mats, C, y, rank, world, dist)
    res = []
    for reml in (True, False):
        np.random.seed(4)
        res.append(ev.evaluate(np.log([0.45, 0.5]), reml=reml, sim_num=50))
    np.savez(out % rank, nll=np.array([r[0] for r in res]), grad=np.array([r[1] for r in res]), l0=eng.l0,
             nlevels=eng.nlevels, computed=eng.panels_computed, sent=eng.panels_sent, ns=eng.cpu.ns, perm=eng.perm,
             level=eng.level)
    dist.destroy_process_group()


def test_two_rank_distributed_chain_matches_single_process(tmp_path):
    """The giant component's separator chain is computed half by rank 0, half by rank 1 (panel broadcasts over gloo);
    nll and gradient of a REML / ML evaluation must match the single-process oracle (same P, same RNG stream)."""
    import torch.multiprocessing as mp
    from scipy.sparse.csgraph import connected_components
    from oracle import reml_oracle as RO
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_chain_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = [np.load(out % r) for r in range(2)]
    mats, C, y = _chain_problem()
    perm = got[0]["perm"]
    for i, reml in enumerate((True, False)):
        np.random.seed(4)
        nll, grad = RO.evaluate(np.log([0.45, 0.5]), mats, C, y, reml, 50, perm=perm)
        for g in got:  # every rank ends with the same numbers
            assert abs(g["nll"][i] - nll) < 1e-10 * abs(nll)
            assert np.abs(g["grad"][i] - grad).max() < 1e-8 * np.abs(grad).max()
    # the split is real: a chain of >= 4 panels exists, each rank computed only its share of it, every chain panel
    # went over the wire, and the chain lies inside ONE connected component (the largest)
    l0, nlev, ns = int(got[0]["l0"]), int(got[0]["nlevels"]), int(got[0]["ns"])
    nchain = nlev - l0
    assert nchain >= 4
    for r, g in enumerate(got):
        mine = len([j for j in range(nchain) if j % 2 == r])
        assert int(g["computed"]) == 2 * ((ns - nchain) + mine)      # two evaluations
        assert int(g["sent"]) == 2 * nchain
    from scilmm_amd.factor import Symbolic
    sym = Symbolic(mats, upload=False)
    sn_start, level = sym.get("sn_start"), got[0]["level"]
    _, label = connected_components(mats[0], directed=False)
    chain_cols = np.concatenate([np.arange(sn_start[s], sn_start[s + 1]) for s in range(ns) if level[s] >= l0])
    comp = np.unique(label[perm[chain_cols]])
    assert comp.size == 1 and comp[0] == np.bincount(label).argmax()


def test_column_chunks_cover_all_columns():
    from scilmm_amd.dist import column_chunks
    for r in (1, 5, 103, 104):
        for world in (1, 2, 3, 8):
            ch = column_chunks(r, world)
            assert ch[0][0] == 0 and ch[-1][1] == r and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
            assert max(c1 - c0 for c0, c1 in ch) - min(c1 - c0 for c0, c1 in ch) <= 1
