"""Time the device dominance builder (csrc/dominance.hip) against the NumPy restatement on a simulated pedigree.

usage: python tools/dominance_bench.py <individuals> <sparsity_factor> [--oracle]
Prints one JSON line: entries, host-buffer call time (upload + kernel + download), kernel-only time from HIP events
through the device-pointer entry point, and (with --oracle) the oracle's time and bitwise agreement.
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from scilmm_amd import _lib  # noqa: E402
from scilmm_amd.harness import pedigree as H  # noqa: E402


def main():
    n, sf = int(sys.argv[1]), float(sys.argv[2])
    par, _, _ = H.simulate_pedigree(n, sf, 0)
    A = H.ibd_from_parents(par).tocsr()
    A.sort_indices()
    t0 = time.time()
    D = _lib.dominance(A, par)
    t_host = time.time() - t0
    # kernel alone: device buffers through torch, HIP events on the current stream
    dev = torch.device("cuda:0")
    ip = torch.from_numpy(A.indptr.astype(np.int64)).to(dev)
    ix = torch.from_numpy(A.indices.astype(np.int32)).to(dev)
    dv = torch.from_numpy(A.data).to(dev)
    pr = torch.from_numpy(np.ascontiguousarray(par, dtype=np.int32)).to(dev)
    out = torch.empty_like(dv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream()
    ms = []
    for rep in range(3):
        e0.record()
        st = _lib.lib().scilmm_dominance_dev(A.shape[0], ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), pr.data_ptr(), out.data_ptr(),
                                             stream.cuda_stream)
        assert st == 0
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    assert np.array_equal(out.cpu().numpy(), D.data)
    res = {"individuals": n, "n": A.shape[0], "entries": int(A.nnz), "host_call_s": t_host, "kernel_ms": min(ms),
           "entries_per_s_kernel": A.nnz / (min(ms) / 1e3),
           "bytes_per_entry_algorithmic": "12 read (index + value) + 8 written + 4 gathered lookups of 8 B",
           "gb_per_s_streamed": 20.0 * A.nnz / (min(ms) / 1e3) / 1e9}
    if "--oracle" in sys.argv:
        from oracle import oracle as O
        t0 = time.time()
        Do = O.dominance(par, A)
        res["oracle_s"] = time.time() - t0
        D.eliminate_zeros()
        res["bitwise_equal_to_oracle"] = bool(np.array_equal(D.indptr, Do.indptr) and np.array_equal(D.indices, Do.indices) and
                                              np.array_equal(D.data, Do.data))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
