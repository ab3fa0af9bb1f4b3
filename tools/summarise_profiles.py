#!/usr/bin/env python3
"""gpurun_out/<tag>_* (written by tools/collect_profiles.sh on the GPU box) -> profiles/<tag>_*.{csv,json,md}"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def stats(src, dst, title):
    f = glob.glob(os.path.join(G, src, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(P, dst), "w") as o:
        o.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats ({os.path.basename(f)})\n")
        o.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows:
            o.write("%s,%s,%.3f,%.2f,%.2f,%.2f,%s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                    float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    return {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}


bench = stats(f"{tag}_bench_stats", f"{tag}_bench_kernel_stats.csv",
              "python bench.py --no-cpu-baseline --no-files (128 scenes per step, 16 pipelines in flight), 150k/1.5k scenes")
solo = stats(f"{tag}_solo_stats", f"{tag}_single_stream_kernel_stats.csv", "tools/time_scene.py 150000 1500 (one scene at a time, 6 forwards)")

traffic = collections.defaultdict(dict)
for name, src in (("FETCH_SIZE", f"{tag}_pmc_fetch"), ("WRITE_SIZE", f"{tag}_pmc_write")):
    f = glob.glob(os.path.join(G, src, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        traffic[k][name] = sum(v) / len(v)
alias = {"k_edgeconv<2, true>": "k_edgeconv<S2X>", "k_edgeconv<2>": "k_edgeconv<S2X>", "k_edgeconv<0, false>": "k_edgeconv<S1>",
         "k_edgeconv<0>": "k_edgeconv<S1>", "k_edgeconv<1, false>": "k_edgeconv<S1X>", "k_edgeconv<1>": "k_edgeconv<S1X>", "k_cluster_knn_pruned<20>": "k_cluster_knn_pruned", "k_cluster_knn_sorted<20>": "k_cluster_knn_sorted",
         "k_cluster_knn_sorted<20, 1>": "k_cluster_knn_sorted", "k_cluster_knn_sorted<20, 2>": "k_cluster_knn_sorted<2 slices>",
         "k_cluster_knn_sorted<20, 4>": "k_cluster_knn_sorted<4 slices>",
         # layer 2 (unseeded) and layer 3 (seeded) launches of the same kernel: bench.py prices them together
         "k_cluster_knn_sorted<20, 1, false>": "k_cluster_knn_sorted", "k_cluster_knn_sorted<20, 1, true>": "k_cluster_knn_sorted"}
out = {"note": "HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB * 1024 from two separate rocprofv3 --pmc passes over "
               "tools/time_scene.py 150000 1500; on gfx950 FETCH_SIZE can under-count wide coalesced reads by up to 2x "
               "(MI355X_MICROARCH.md, HBM section), so read the fetch side as a lower bound",
       "bytes_per_launch": {}, "fetch_kb": {}, "write_kb": {}}
merged = collections.defaultdict(list)
for k, v in traffic.items():
    merged[alias.get(k, k)].append(v)
for name, vs in merged.items():                      # several instantiations under one name: the mean per launch
    f_ = sum(v.get("FETCH_SIZE", 0) for v in vs) / len(vs)
    w_ = sum(v.get("WRITE_SIZE", 0) for v in vs) / len(vs)
    out["bytes_per_launch"][name] = int((f_ + w_) * 1024)
    out["fetch_kb"][name] = round(f_, 1)
    out["write_kb"][name] = round(w_, 1)
json.dump(out, open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
line = [l for l in open(os.path.join(G, f"{tag}_bench_stats.log")).read().splitlines() if l.startswith('{"metric"')][-1]
with open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w") as o:
    o.write(line + "\n")
sq = os.path.join(G, f"{tag}_pmc_sq.txt")
if os.path.exists(sq):
    lines = [l for l in open(sq).read().splitlines() if " per wave" in l or "| per wave" in l]
    with open(os.path.join(P, f"{tag}_sq_counters.txt"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                "SQ_ACTIVE_INST_ANY -- python3 tools/time_scene.py 150000 1500\n# averages per launch; wave-cycles = 4 x SQ_WAVE_CYCLES / waves\n")
        o.write("\n".join(lines) + "\n")
print("top kernels (bench, avg us | solo avg us):")
for k in list(bench)[:8]:
    print("  %-28s %9.1f | %9.1f" % (k, bench[k], solo.get(k, float("nan"))))


# ---- stress config (BASELINE.json configs[4]): 500k points / 5k segments / 20-NN, rocprof HBM/MFMA roofline report
sf = glob.glob(os.path.join(G, f"{tag}_stress_stats", "**", "*kernel_stats.csv"), recursive=True)
if sf:
    st = stats(f"{tag}_stress_stats", f"{tag}_stress_500k_kernel_stats.csv", "tools/time_scene.py 500000 5000 (one scene at a time, 6 forwards)")
    N, S, k = 500000.0, 5000.0, 20.0
    E0 = 3.57 * N
    model = {   # kernel: (bound, algorithmic units per launch, note)   -- DESIGN.md section 4
        "k_edgeconv<2, true>": ("mfma", 2 * k * N * (18 * 64 + 64 * 64), "MLP3 conv1'->conv2 + BN2 statistics + max (S2X)"),
        "k_edge_moments": ("hbm", (80 + 48 * k) * N, "MLP3 inner-BN statistics from edge-feature moments (gather-latency-bound)"),
        "k_edgeconv<1, false>": ("mfma", 2 * k * N * 18 * 64, "MLP2 conv + BN statistics + max (S1X)"),
        "k_cluster_knn_sorted<20, 1, false>": ("hbm", 96 * N, "in-cluster kNN-20, layer 2 (VALU/latency-bound; HBM is the nominal roof)"),
        "k_cluster_knn_sorted<20, 1, true>": ("hbm", 96 * N, "in-cluster kNN-20, layer 3, seeded from layer 2"),
        "k_segment_max64": ("hbm", 260 * N, "per-cluster max of [N,64]"),
        "k_export": ("hbm", 60 * N, "14 label vectors gather"),
        "k_mark_pairs": ("hbm", 16 * E0, "mesh-edge contraction (bitmap)"),
        "k_center_write": ("hbm", 92 * N, "per-cluster centring, writes x9m + kNN operand"),
        "k_gather_members": ("hbm", 12 * N, "member lists"),
    }
    with open(os.path.join(P, f"{tag}_stress_500k_report.md"), "w") as o:
        o.write("# Stress scene 500k points / 5k segments / 20-NN graph, 1x MI355X (BASELINE.json configs[4])\n\n")
        wall = [l for l in open(os.path.join(G, f"{tag}_stress_stats.log")).read().splitlines() if l.startswith("iter 3")]
        o.write("`rocprofv3 --kernel-trace --stats -- python3 tools/time_scene.py 500000 5000`; " + (wall[0] if wall else "") + "\n\n")
        o.write("Parity at this size: `tests/test_gpu_scene.py::test_stress_500k_matches_reference_and_oracle_digests` (14 label vectors == reference capture == oracle).\n\n")
        o.write("| kernel | avg us / launch | algorithmic work / launch | achieved | roof | fraction |\n|---|---|---|---|---|---|\n")
        for kname, (bound, units, note) in model.items():
            if kname not in st:
                continue
            us = st[kname]
            if bound == "mfma":
                ach = units / (us * 1e-6) / 1e12
                o.write(f"| `{kname}` ({note}) | {us:.1f} | {units/1e9:.1f} GFLOP | {ach:.1f} TFLOP/s | 157.3 TFLOP/s fp32 MFMA | {ach/157.3:.3f} |\n")
            else:
                ach = units / (us * 1e-6) / 1e9
                o.write(f"| `{kname}` ({note}) | {us:.1f} | {units/1e6:.1f} MB | {ach:.0f} GB/s | 8000 GB/s HBM | {ach/8000:.4f} |\n")
        o.write("\nWhole scene: the path is latency/compute-bound, not HBM-bound (SURVEY.md 8d): ~470 MB of algorithmic HBM traffic per "
                "500k scene in ~10 ms = 47 GB/s = 0.6 % of the HBM roof; the dense contraction (131 GFLOP per scene) runs on fp32 MFMA.\n")
