"""Multi-rank path of the HIP engine, rehearsed on ONE GPU: two processes share the card and exchange the chain
panels through torch.distributed/gloo (RCCL refuses two ranks on one device; the engine only sees a callback, so
the code path -- ownership filter of the plan, event ordering around the broadcasts, replicated factor, column-split
solves -- is the one that runs with backend "nccl" on a multi-GPU node).  The file name sorts last on purpose: a
failure here must not take the single-GPU parity tests with it."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

from tests.helpers import rel_err, small_pedigree

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    A, sex = small_pedigree(20000, 0.01, 3)
    n = A.shape[0]
    rng = np.random.default_rng(2)
    y = rng.standard_normal(n)
    C = np.stack([(sex - sex.mean()) / sex.std(), np.ones(n)], axis=1)
    return [A, sp.eye(n).tocsr()], C, y


def _worker(rank, world, port, out, env):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)  # a stuck collective must not leave a process on the GPU
    os.environ.update(env)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scilmm_amd.dist import DistributedEvaluator, HipChainEngine
    mats, C, y = _problem()
    eng = HipChainEngine(mats, rank, world, dist, "cuda:0")
    ev = DistributedEvaluator(eng, mats, C, y, rank, world, dist, device="cpu")
    np.random.seed(4)
    nll, grad = ev.evaluate(np.log([0.45, 0.5]), reml=True, sim_num=50)
    ld1 = eng.logdet()
    rng = np.random.default_rng(0)
    B = rng.standard_normal((y.size, 7))
    X1 = eng.solve_local(B)
    eng.factorize([0.3, 0.7])  # a second factorization on the same handle (stale panels must not survive)
    np.savez(out % rank, nll=nll, grad=grad, logdet=ld1, X=X1, logdet2=eng.logdet(), X2=eng.solve_local(B),
             Z2=eng.lmul_local(B), perm=eng.P())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("env", [{}, {"SCILMM_TUNING": "1", "SCILMM_DENSE": "1", "SCILMM_OUTSIDE": "1"}],
                         ids=["default", "dense+outside"])
def test_two_ranks_one_gpu_match_single_process(tmp_path, env):
    """env 2 forces, at this small size, the schedule the 300k / 1M configurations get by default on every rank of a
    multi-GPU run: dense-tail kernel + atomic prelude -> tail contributions, restricted to the panels a rank owns."""
    import torch.multiprocessing as mp
    from oracle import reml_oracle as RO
    from scilmm_amd.factor import Symbolic
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, env), nprocs=2, join=True)
    got = [np.load(out % r) for r in range(2)]
    mats, C, y = _problem()
    sym = Symbolic(mats)
    assert np.array_equal(got[0]["perm"], sym.P()) and np.array_equal(got[1]["perm"], sym.P())
    f = sym.factorize([0.45, 0.5])
    rng = np.random.default_rng(0)
    B = rng.standard_normal((y.size, 7))
    X = f(B)
    np.random.seed(4)
    nll, grad = RO.evaluate(np.log([0.45, 0.5]), mats, C, y, True, 50, perm=sym.P())
    f2 = sym.factorize([0.3, 0.7])
    for g in got:
        assert abs(g["logdet"] - f.logdet()) < 1e-11 * abs(f.logdet())
        assert rel_err(g["X"], X) < 1e-10
        assert abs(g["nll"] - nll) < 1e-10 * abs(nll)
        assert rel_err(g["grad"], grad) < 1e-7
        assert abs(g["logdet2"] - f2.logdet()) < 1e-11 * abs(f2.logdet())
        assert rel_err(g["X2"], f2(B)) < 1e-10
        assert rel_err(g["Z2"], f2.lmul(B)) < 1e-10
