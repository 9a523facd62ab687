"""Debug helper: 2 ranks on one GPU (gloo), distributed tail with front precision 64 / 32 (k_dense32) / 32 + shadow (k_dense_h)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SCILMM_TUNING"] = "1"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scilmm_amd.dist import HipChainEngine
    from tests.helpers import small_pedigree
    A, sex = small_pedigree(20000, 0.01, 1)
    n = A.shape[0]
    mats = [A, sp.eye(n).tocsr()]
    B = np.random.default_rng(0).standard_normal((n, 7))
    res = {}
    for tag, bits, shadow in (("fp64", 64, "1"), ("k_dense32", 32, "0"), ("k_dense_h", 32, "1")):
        os.environ["SCILMM_SHADOW"] = shadow
        eng = HipChainEngine(mats, rank, world, dist, "cuda:0")
        eng.sym.set_front_precision(bits)
        eng.factorize([0.45, 0.5])
        from scilmm_amd.dist import tail_layout
        info = eng.sym.info()
        owner, loff, params = tail_layout(eng.sym._h, info.nsuper, rank, world)
        Lloc = eng._bufs[0].cpu().numpy().copy()
        res[tag] = (eng.logdet(), eng.solve(B), Lloc, np.asarray(loff), int(params[0]), info.nsuper, np.asarray(owner))
        del eng
    ld0, X0, L0, loff, first, nsuper, owner = res["fp64"]
    for tag in ("k_dense32", "k_dense_h"):
        ld, X, L1 = res[tag][:3]
        if rank == 0:
            print(tag, "logdet diff", abs(ld - ld0), "X rel", np.abs(X - X0).max() / np.abs(X0).max(), flush=True)
        bad = []
        for f in range(first, nsuper):
            if owner[f] != rank:
                continue
            a, b = int(loff[f]), None
            seg0, seg1 = L0[a:a + 200000], L1[a:a + 200000]
            bad.append((f - first, float(np.abs(seg0 - seg1).max())))
        print("rank", rank, tag, "own panels (jj, max |dL| in the first 200k entries):", [(j, "%.1e" % d) for j, d in bad], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)
