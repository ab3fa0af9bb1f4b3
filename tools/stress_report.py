#!/usr/bin/env python3
"""gpurun_out/<tag>_stress_{stats,fetch,write} (tools/stress_500k.sh) -> profiles/<tag>_stress_500k_report.md + _kernel_stats.csv

Per kernel of ONE forward of the stress scene (BASELINE.json configs[4]): launches per forward, us per launch, HBM bytes per launch
from the PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; KB units), the achieved GB/s
against the 8 TB/s HBM roof and, for the EdgeConv launches, the executed 16-bit MFMA TFLOP/s against the 2.5 PF dense peak.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag, N, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(n):
    return re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


stats = newest(os.path.join(G, f"{tag}_stress_stats", "**", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
log = open(os.path.join(G, f"{tag}_stress_stats.log")).read()
line = [l for l in log.splitlines() if l.startswith('{"points"')]
meta = json.loads(line[-1]) if line else {}
iters = len([l for l in log.splitlines() if l.startswith("iter ")]) or 1


def pmc(name, counter):
    f = newest(os.path.join(G, f"{tag}_stress_{name}", "**", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
valu = pmc("valu", "SQ_INSTS_VALU")            # wave-instructions per launch (the kNN kernels' roof is VALU issue: 512 G wave-instructions/s)
# executed MFMA flop per edge row (bench.py: CONV1 12 MFMAs of 32x32x16 per 32 rows, conv2 three fp16 products of 64x64)
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def edgeconv_mode(k):
    """1 (MLP2) / 2 (MLP3) for any EdgeConv launch -- k_edgeconv<MODE, ...>, k_edgeconv_h<MODE>, k_edgeconv_b / _hb -- else 0"""
    m = re.match(r"k_edgeconv(_h|_b|_hb)?<(\d)", k)
    return int(m.group(2)) if m else 0


def flops_of(k):
    """(algorithmic fp32-contraction flop, executed 16-bit MFMA flop) of one launch over the N-point scene"""
    mode = edgeconv_mode(k)
    if mode == 1:
        return 2.0 * 20 * N * (18 * 64), bench.S1X_EXECUTED_FLOP_PER_ROW * 20.0 * N
    if mode == 2:
        return 2.0 * 20 * N * (18 * 64 + 64 * 64), bench.S2X_EXECUTED_FLOP_PER_ROW * 20.0 * N
    return None


tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = []
with open(os.path.join(P, f"{tag}_stress_500k_kernel_stats.csv"), "w") as o:
    o.write(f"# python tools/time_scene.py {N} {S}: one scene at a time through sg_pipeline_forward, {iters} forwards\n")
    w = csv.writer(o)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                    "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]])
for r in rows:
    k = short(r["Name"])
    us = float(r["AverageNs"]) / 1e3
    per_fwd = int(r["Calls"]) / iters
    b = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
    gbs = b / (us * 1e-6) / 1e9 if us > 0 and b > 0 else 0.0
    fl = flops_of(k)
    tf = (fl[0] / (us * 1e-6) / 1e12, fl[1] / (us * 1e-6) / 1e12) if fl and us > 0 else None
    vfrac = valu[k] / (us * 1e-6) / (bench.VALU_PEAK_GINST * 1e9) if k in valu and us > 0 and k.startswith("k_cluster_knn") else None
    out.append((k, per_fwd, us, us * per_fwd, b, gbs, tf, 100 * float(r["TotalDurationNs"]) / tot, vfrac))
with open(os.path.join(P, f"{tag}_stress_500k_report.md"), "w") as o:
    o.write(f"# Stress scene {N} points / {S} segments / 20-NN graph on 1x MI355X (BASELINE.json configs[4])\n\n")
    o.write(f"`tools/stress_500k.sh {tag}`: `rocprofv3 --kernel-trace --stats -- python3 tools/time_scene.py {N} {S}` (one scene at a time through the single-scene "
            "pipeline, default stream) + two separate `--pmc` passes (`FETCH_SIZE`, `WRITE_SIZE`).  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB "
            "(gfx950 correction of `MI355X_MICROARCH.md`); GB/s against the 8,000 GB/s HBM roof.  EdgeConv launches: ALGORITHMIC TFLOP/s "
            "(2 x 20 x N x (18 x 64 [+ 64 x 64]) fp32-contraction flop / time) against the 2,500 TF dense 16-bit MFMA peak, and beside it the flop the "
            "two-piece fp16 emulation EXECUTES.  kNN launches: `SQ_INSTS_VALU` (a fourth pass) / time against the measured VALU issue roof of "
            "%.0f G wave-instructions/s.\n\n" % bench.VALU_PEAK_GINST)
    if meta:
        o.write(f"Scene: E0 = {meta.get('E0')} mesh edges, V = {meta.get('V')} raw vertices, cluster trace {meta.get('trace')}.  Wall per forward "
                f"(kernels + serial host grouping + D2H of the 14 label vectors): median {meta.get('wall_ms_median')} ms, min {meta.get('wall_ms_min')} ms; "
                f"device time summed over the kernels below: {sum(x[3] for x in out) / 1e3:.2f} ms.\n\n")
    o.write("| kernel | launches / forward | us / launch | us / forward | % | HBM MB / launch | GB/s | frac of 8 TB/s | MFMA TFLOP/s algorithmic (frac of 2.5 PF); executed | VALU issue frac |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for k, n_, us, usf, b, gbs, tf, pct, vfrac in out:
        if pct < 0.05:
            continue
        o.write("| `%s` | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |\n" % (
            k, n_, us, usf, pct, ("%.2f" % (b / 1e6)) if b else "-", ("%.0f" % gbs) if gbs else "-", ("%.3f" % (gbs / 8000.0)) if gbs else "-",
            ("%.0f (%.3f); %.0f (%.3f)" % (tf[0], tf[0] / 2500.0, tf[1], tf[1] / 2500.0)) if tf else "-", ("%.2f" % vfrac) if vfrac else "-"))
    if meta.get("stage_ms"):
        o.write("\nPipeline stage times of the last forward (HIP events, ms): `%s`\n" % json.dumps(meta["stage_ms"]))
print(open(os.path.join(P, f"{tag}_stress_500k_report.md")).read()[:3000])
