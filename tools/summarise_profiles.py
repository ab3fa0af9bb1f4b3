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
              "python bench.py --no-cpu-baseline --no-files (8 scenes x 6 steps, 4 pipelines in flight), 150k/1.5k scenes")
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
alias = {"k_edgeconv<3>": "k_edgeconv<FINAL2>", "k_edgeconv<2>": "k_edgeconv<STATS2>", "k_edgeconv<0>": "k_edgeconv<STATS1>",
         "k_edgeconv<1>": "k_edgeconv<FINAL1>", "k_cluster_knn_pruned<20>": "k_cluster_knn_pruned"}
out = {"note": "HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB * 1024 from two separate rocprofv3 --pmc passes over "
               "tools/time_scene.py 150000 1500; on gfx950 FETCH_SIZE can under-count wide coalesced reads by up to 2x "
               "(MI355X_MICROARCH.md, HBM section), so read the fetch side as a lower bound",
       "bytes_per_launch": {}, "fetch_kb": {}, "write_kb": {}}
for k, v in traffic.items():
    name = alias.get(k, k)
    out["bytes_per_launch"][name] = int((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024)
    out["fetch_kb"][name] = round(v.get("FETCH_SIZE", 0), 1)
    out["write_kb"][name] = round(v.get("WRITE_SIZE", 0), 1)
json.dump(out, open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
line = [l for l in open(os.path.join(G, f"{tag}_bench_stats.log")).read().splitlines() if l.startswith('{"metric"')][-1]
with open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w") as o:
    o.write(line + "\n")
print("top kernels (bench, avg us | solo avg us):")
for k in list(bench)[:8]:
    print("  %-28s %9.1f | %9.1f" % (k, bench[k], solo.get(k, float("nan"))))
