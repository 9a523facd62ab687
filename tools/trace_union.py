"""Interval union of one kernel's launches in a rocprofv3 kernel trace.

    python tools/trace_union.py <kernel_trace.csv> [name-regex, default k_dense_b] [--by-step N]

rocprofv3 --kernel-trace writes one row per dispatch with Start_Timestamp / End_Timestamp (ns).  When two launches of
the dominant kernel overlap on the engine's two look-ahead streams, the SUM of their durations (what --stats averages)
counts the shared time twice; the UNION of the [start, end] intervals is the time during which at least one launch of
that kernel was running -- the denominator of bench.py's `roofline.achieved_over_busy_time`.  Prints launches, summed
duration, union, and the overlap factor (sum / union)."""
import csv
import re
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    path = sys.argv[1]
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "k_dense_b")
    iv = []
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh)
        name_key = next(k for k in rd.fieldnames if k.lower() in ("kernel_name", "name"))
        s_key = next(k for k in rd.fieldnames if k.lower().startswith("start"))
        e_key = next(k for k in rd.fieldnames if k.lower().startswith("end"))
        for row in rd:
            if pat.search(row[name_key]):
                iv.append((int(row[s_key]), int(row[e_key])))
    if not iv:
        raise SystemExit("no dispatch matches %r" % pat.pattern)
    iv.sort()
    total = sum(b - a for a, b in iv)
    union, cur0, cur1 = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur1:
            union += cur1 - cur0
            cur0, cur1 = a, b
        else:
            cur1 = max(cur1, b)
    union += cur1 - cur0
    print("kernel /%s/: %d launches, summed duration %.3f s (avg %.3f ms), union of the intervals %.3f s, sum / union = %.2f"
          % (pat.pattern, len(iv), total / 1e9, total / 1e6 / len(iv), union / 1e9, total / union))


if __name__ == "__main__":
    main()
