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
