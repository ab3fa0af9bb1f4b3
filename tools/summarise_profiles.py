#!/usr/bin/env python3
"""gpurun_out/<tag>_* (written on the GPU box by tools/prof_engine.sh and tools/pmc_engine.sh) -> profiles/<tag>_*

    python tools/summarise_profiles.py r02

  profiles/<tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the bench's default engine shape
  profiles/<tag>_solo_batched_kernel_stats.csv the same with ONE group of 8 scenes (batched launches, nothing else on the GPU)
  profiles/<tag>_pmc_kernels.json              per kernel and SCENE-launch (a batched launch / the scenes in it): VALU instructions,
                                               MFMA busy share, wave occupancy, HBM bytes (FETCH_SIZE, WRITE_SIZE: separate PMC passes)
  profiles/<tag>_sq_counters.txt               the SQ counters as a table
bench.py reads <tag>_pmc_kernels.json for `roofline.traffic` and the kNN's VALU roof.
"""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
SIMDS = 1024


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def stats(src_dir, dst, title):
    fs = sorted(glob.glob(os.path.join(G, src_dir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        return None
    fs = fs[-1:]                                               # the newest run merged into gpurun_out/
    rows = list(csv.DictReader(open(fs[0])))
    with open(os.path.join(P, dst), "w") as o:
        o.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats ({os.path.basename(fs[0])})\n")
        w = csv.writer(o)                                      # kernel names carry commas (template arguments): quoted
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]])
    return {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}


bench = stats("prof_bench", f"{tag}_bench_kernel_stats.csv", "python bench.py --no-cpu-baseline --no-files (default engine shape), 150k/1.5k scenes")
solo = stats("prof_solo8", f"{tag}_solo_batched_kernel_stats.csv", "python bench.py --groups 1 --per-group 8 --no-cpu-baseline --no-files: one group, 8 scenes per batched launch, nothing else on the GPU")

train = stats("prof_train", f"{tag}_train_kernel_stats.csv", "python tools/time_train.py --steps 6: the training step (forward with tape + loss + backward + SGD), 150k/1.5k scenes, one scene per step")

passes = {}
for name in ("sq", "mfma", "fetch", "write"):
    p = os.path.join(G, f"{tag}_pmc_{name}.json")
    if os.path.exists(p):
        passes[name] = json.load(open(p))
if passes:
    eng = next(iter(passes.values()))["engine"]
    B = int(eng.split(" x ")[1].split()[0])
    alias = {"k_edgeconv_b<2, true>": "k_edgeconv<S2X>", "k_edgeconv_b<1, false>": "k_edgeconv<S1X>",
             "k_edgeconv_hb<2>": "k_edgeconv<S2X>", "k_edgeconv_hb<1>": "k_edgeconv<S1X>",
             "k_cluster_knn_sorted_b<20, 1, false>": "k_cluster_knn_sorted<unseeded>", "k_cluster_knn_sorted_b<20, 1, true>": "k_cluster_knn_sorted<seeded>"}
    out = {"configuration": f"solo batched: {eng}, python bench.py --steps 2 --warmup 1 under rocprofv3 --pmc (one counter set per pass, --kernel-trace only)",
           "scenes_per_launch": B,
           "note": "per SCENE-launch = per batched launch / scenes per launch.  hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024: on gfx950 FETCH_SIZE reports half "
                   "the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); hbm_bytes_uncorrected uses FETCH_SIZE as reported.  MFMA busy share = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); mean waves per SIMD = 4 x SQ_WAVE_CYCLES / 1024 / kernel cycles.",
           "valu_insts_per_scene_launch": {}, "hbm_bytes_per_scene_launch": {}, "hbm_bytes_uncorrected_per_scene_launch": {}, "mfma_busy_share": {},
           "mean_waves_per_simd": {}, "issue_share": {}, "per_kernel_raw": {}}
    sq, mf = passes.get("sq", {}).get("kernels", {}), passes.get("mfma", {}).get("kernels", {})
    fe, wr = passes.get("fetch", {}).get("kernels", {}), passes.get("write", {}).get("kernels", {})
    lines = []
    for k in sorted(set(sq) | set(mf) | set(fe) | set(wr)):
        if not k.endswith("_b") and "_b<" not in k and "_hb<" not in k:
            continue                                           # batched kernels only (the single-pipeline parity check runs a few unbatched ones)
        name = alias.get(k, k)
        raw = {}
        for src in (sq, mf, fe, wr):
            raw.update({n: v for n, v in src.get(k, {}).items() if n != "launches"})
        out["per_kernel_raw"][name] = {n: round(v, 1) for n, v in raw.items()}
        if "SQ_INSTS_VALU" in raw:
            out["valu_insts_per_scene_launch"][name] = round(raw["SQ_INSTS_VALU"] / B)
        if "FETCH_SIZE" in raw or "WRITE_SIZE" in raw:
            out["hbm_bytes_per_scene_launch"][name] = int((2 * raw.get("FETCH_SIZE", 0) + raw.get("WRITE_SIZE", 0)) * 1024 / B)
            out["hbm_bytes_uncorrected_per_scene_launch"][name] = int((raw.get("FETCH_SIZE", 0) + raw.get("WRITE_SIZE", 0)) * 1024 / B)
        cyc = raw.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in raw:
            out["mfma_busy_share"][name] = round(raw["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / cyc, 4)
        # SQ pass and MFMA pass are different runs: kernel cycles for the occupancy figure come from the MFMA pass's GRBM_GUI_ACTIVE
        if cyc and "SQ_WAVE_CYCLES" in raw:
            out["mean_waves_per_simd"][name] = round(4.0 * raw["SQ_WAVE_CYCLES"] / SIMDS / cyc, 2)
        if raw.get("SQ_WAVE_CYCLES"):
            out["issue_share"][name] = {"issuing": round(raw.get("SQ_ACTIVE_INST_ANY", 0) / raw["SQ_WAVE_CYCLES"], 3),
                                        "waiting_s_waitcnt": round(raw.get("SQ_WAIT_ANY", 0) / raw["SQ_WAVE_CYCLES"], 3),
                                        "issue_stall": round(raw.get("SQ_WAIT_INST_ANY", 0) / raw["SQ_WAVE_CYCLES"], 3)}
            w = max(raw.get("SQ_WAVES", 1), 1)
            lines.append("%-40s waves/launch %7d | per wave: VALU %7.0f SALU %6.0f LDS %6.0f | issuing %4.1f%% s_waitcnt %4.1f%% issue-stall %4.1f%%" % (
                name[:40], w, raw.get("SQ_INSTS_VALU", 0) / w, raw.get("SQ_INSTS_SALU", 0) / w, raw.get("SQ_INSTS_LDS", 0) / w,
                100 * raw.get("SQ_ACTIVE_INST_ANY", 0) / raw["SQ_WAVE_CYCLES"], 100 * raw.get("SQ_WAIT_ANY", 0) / raw["SQ_WAVE_CYCLES"],
                100 * raw.get("SQ_WAIT_INST_ANY", 0) / raw["SQ_WAVE_CYCLES"]))
    json.dump(out, open(os.path.join(P, f"{tag}_pmc_kernels.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(P, f"{tag}_sq_counters.txt"), "w") as o:
        o.write(f"# {out['configuration']}\n# SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY, averages per batched launch\n")
        o.write("\n".join(lines) + "\n")
    print("pmc: %d kernels" % len(out["per_kernel_raw"]))
for name, d in (("bench", bench), ("solo batched", solo), ("training step", train)):
    if d:
        print(name, "top kernels (avg us per launch):")
        for k in list(d)[:8]:
            print("   %-44s %9.1f" % (k, d[k]))
log = os.path.join(G, "prof_bench.log")
if os.path.exists(log):
    ls = [l for l in open(log).read().splitlines() if l.startswith('{"metric"')]
    if ls:
        open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w").write(ls[-1] + "\n")
