"""HBM traffic of the update kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) -> profiles/r<round>_traffic_<workload>.json, the file bench.py quotes as `roofline.traffic`.

usage: traffic_json.py <workload> <fetch_counter_collection.csv> <write_counter_collection.csv> <n_factorizations> <out.json>

Corrections: FETCH_SIZE x2 on gfx950 (the counter tallies 128-B requests at 64 B; calibrated in round 1 on k_assemble's
known byte count for this code's 8-B-per-lane coalesced loads), WRITE_SIZE as read; both counters are in KB (1000 B).
"""
import json
import sys

import pandas as pd


def by_kernel(path):
    d = pd.read_csv(path)
    name = "Kernel_Name" if "Kernel_Name" in d.columns else "Kernel Name"
    val = "Counter_Value" if "Counter_Value" in d.columns else "Counter Value"
    d["k"] = d[name].str.extract(r"(k_[a-z_0-9]+)")[0].fillna(d[name].str.slice(0, 40))
    g = d.groupby("k").agg(value=(val, "sum"), dispatches=(val, "size"))
    return g


def main():
    workload, fpath, wpath, nfact, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    f, w = by_kernel(fpath), by_kernel(wpath)
    upd = [k for k in f.index if k in ("k_dense", "k_dense_g", "k_dense_a", "k_dense_b", "k_dense32", "k_update2", "k_update3", "k_update")]
    fetch_kb = float(f.loc[upd, "value"].sum()) / nfact
    write_kb = float(w.loc[[k for k in upd if k in w.index], "value"].sum()) / nfact
    launches = float(f.loc[upd, "dispatches"].sum()) / nfact
    total = (2.0 * fetch_kb + write_kb) * 1000.0
    res = {
        "kernels": upd, "workload": workload,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --workload %s --steps 1 "
                  "--warmup 0 --no-cpu-baseline` (%d factorizations per pass)" % (workload, nfact),
        "fetch_kb_per_factorization_raw": fetch_kb, "write_kb_per_factorization_raw": write_kb,
        "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request; calibrated in round 1 on k_assemble), WRITE_SIZE as read; KB = 1000 B",
        "bytes_per_factorization": total, "launches_per_factorization": launches,
        "bytes_per_launch": total / max(launches, 1.0),
        "per_kernel_fetch_kb_raw": {k: float(f.loc[k, "value"]) / nfact for k in f.index},
        "per_kernel_write_kb_raw": {k: float(w.loc[k, "value"]) / nfact for k in w.index},
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("bytes_per_factorization", "launches_per_factorization", "bytes_per_launch")}))


if __name__ == "__main__":
    main()
