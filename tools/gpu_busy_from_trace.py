#!/usr/bin/env python3
"""How busy is the GPU inside the bench's timed region?  Reads a rocprofv3 kernel trace (csv) and a marker trace of `bench.py --profile`:
the union of the kernel intervals inside the `timed_region` range / the range's length = the share of time in which AT LEAST ONE kernel was running;
the sum of the durations / the length = the average number of kernels in flight.  (A share well below 1 would mean the engine, not the GPU, is the bound.)

    python3 tools/gpu_busy_from_trace.py <dir with *_kernel_trace.csv and *_marker_api_trace.csv>
"""
import csv
import glob
import os
import sys


def main(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    mt = glob.glob(os.path.join(d, "**", "*marker_api_trace.csv"), recursive=True)[0]
    t0 = t1 = None
    for r in csv.DictReader(open(mt)):
        if r.get("Function", r.get("Name", "")) == "timed_region" or "timed_region" in str(r):
            t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            break
    iv = []
    names = {}
    for r in csv.DictReader(open(kt)):
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is not None and (b <= t0 or a >= t1):
            continue
        a, b = max(a, t0 or a), min(b, t1 or b)
        iv.append((a, b))
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
        names[n] = names.get(n, 0) + (b - a)
    iv.sort()
    union, cur_a, cur_b, gaps = 0, None, None, []
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                union += cur_b - cur_a
                gaps.append(a - cur_b)
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += (cur_b - cur_a) if cur_b is not None else 0
    span = (t1 - t0) if t0 is not None else (iv[-1][1] - iv[0][0])
    total = sum(b - a for a, b in iv)
    print("timed region %.3f ms, %d kernel launches inside" % (span / 1e6, len(iv)))
    print("at least one kernel running: %.1f %% of the region; kernels in flight on average: %.2f" % (100.0 * union / span, total / span))
    gaps.sort(reverse=True)
    print("idle gaps: %d, total %.3f ms, the ten longest (us): %s" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3, 1) for g in gaps[:10]]))
    top = sorted(names.items(), key=lambda kv: -kv[1])[:8]
    print("kernel time / region length (sums overlap): " + ", ".join("%s %.2f" % (k[:28], v / span) for k, v in top))


if __name__ == "__main__":
    main(sys.argv[1])
