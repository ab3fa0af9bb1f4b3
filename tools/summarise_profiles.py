#!/usr/bin/env python3
"""gpurun_out/<tag>_* (written on the GPU box by tools/prof_engine.sh and tools/pmc_engine.sh) -> profiles/<tag>_*

    python tools/summarise_profiles.py r06

  profiles/<tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the bench's default engine shape
  profiles/<tag>_solo_batched_kernel_stats.csv the same with ONE group of 8 scenes (batched launches, nothing else on the GPU)
  profiles/<tag>_pmc_kernels.json              per kernel and SCENE-launch (a batched launch / the scenes in it): VALU instructions,
                                               MFMA busy share, wave occupancy, HBM bytes (FETCH_SIZE, WRITE_SIZE: separate PMC passes)
  profiles/<tag>_sq_counters.txt               the SQ counters as a table
  profiles/<tag>_kernel_table.md               DESIGN.md section 4's per-kernel table, derived from the two files above (nothing typed by hand)
bench.py reads <tag>_pmc_kernels.json for `roofline.traffic` and the kNN's VALU roof.

Round 6: a summary is only worth its per-launch averages if every launch it averages is a FULL launch (round 5's profiled process also ran the
out-of-step parity leg, which launches the same batched kernels with one scene each: every average was diluted by 7.3 %).  `check_summary`
holds a summary to the launch model -- SQ_WAVES(k_mlp1_apply_b) = B x S, SQ_INSTS_MFMA(k_edgeconv<S2X>) = B x ceil(N / 32) x 570, SQ_WAVES of the
persistent EdgeConv grids = 2,048, the solo CSV's min / max of the dominant kernels within a band of their average -- and this script exits non-zero
(after writing the files, with `checks.ok = false` inside) when it does not hold.  tests/test_profiles.py runs the same check on the committed files.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
SIMDS = 1024
# MFMA instructions of one 32-row tile of the hand-scheduled MLP3 launch: 20 slots x (4 conv1 + 24 conv2) + 10 for the x_i half / base
S2X_MFMA_PER_TILE = 570
ALIAS = {"k_edgeconv_b<2, true>": "k_edgeconv<S2X>", "k_edgeconv_b<1, false>": "k_edgeconv<S1X>",
         "k_edgeconv_hb<2>": "k_edgeconv<S2X>", "k_edgeconv_hb<1>": "k_edgeconv<S1X>",
         "k_cluster_knn_sorted_b<20, 1, false>": "k_cluster_knn_sorted<unseeded>", "k_cluster_knn_sorted_b<20, 1, true>": "k_cluster_knn_sorted<seeded>"}


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def read_stats_csv(path):
    """rows of a committed profiles/*_kernel_stats.csv (our format) as dicts with float fields"""
    rows = []
    with open(path) as f:
        lines = [ln for ln in f if not ln.startswith("#")]
    for r in csv.DictReader(lines):
        rows.append({"kernel": r["kernel"], "calls": int(r["calls"]), "avg_us": float(r["avg_us"]), "min_us": float(r["min_us"]),
                     "max_us": float(r["max_us"]), "total_ms": float(r["total_ms"])})
    return rows


def check_summary(pmc, solo_rows, points=150000, segments=1500, band=0.25):
    """-> (ok, [problem strings]).  `pmc` = a <tag>_pmc_kernels.json dict (or None), `solo_rows` = read_stats_csv of the solo CSV (or None)."""
    problems = []
    if pmc:
        B = int(pmc.get("scenes_per_launch", 8))
        points = int(pmc.get("points", points))
        segments = int(pmc.get("segments", segments))
        raw = pmc.get("per_kernel_raw", {})

        def expect(kernel, counter, want, what):
            v = raw.get(kernel, {}).get(counter)
            if v is None:
                problems.append(f"{kernel}: no {counter} in the summary")
            elif abs(v - want) > 1e-6 * want:
                problems.append(f"{kernel}: {counter} = {v:,.1f} per launch, a launch of {B} full scenes has {want:,} ({what}): the summary averages launches of "
                                f"different sizes (ratio {v / want:.4f})")
        expect("k_mlp1_apply_b", "SQ_WAVES", B * segments, "one wave per segment")
        tiles = (points + 31) // 32
        expect("k_edgeconv<S2X>", "SQ_INSTS_MFMA", B * tiles * S2X_MFMA_PER_TILE, f"{tiles} tiles x {S2X_MFMA_PER_TILE} MFMAs per scene")
        expect("k_edgeconv<S2X>", "SQ_WAVES", 2048, "persistent grid: two waves per SIMD")
        # min / max per launch, when the collection kept them: kernels with ONE call site per group-step and a grid that depends on nothing but
        # the scenes' N and S launch the same number of waves every time (others -- k_edge_distance_b x 3, k_cluster_affine_b x 2, the kNN's tile
        # counts -- legitimately differ between their call sites and batches)
        for k in ("k_mlp1_apply_b", "k_edgeconv<S2X>", "k_edgeconv<S1X>", "k_mlp1_knn_moments_b", "k_fps_sample_b<64>"):
            c = raw.get(k, {})
            for n in ("SQ_WAVES", "SQ_INSTS_MFMA"):
                lo, hi, av = c.get(n + "__min"), c.get(n + "__max"), c.get(n)
                if lo is None or hi is None or not av:
                    continue
                if (hi - lo) > 1e-6 * av:
                    problems.append(f"{k}: {n} per launch ranges {lo:,.0f} .. {hi:,.0f} (average {av:,.1f}): launches of different sizes")
    if solo_rows is not None:
        by = {r["kernel"]: r for r in solo_rows}
        for k in ("k_edgeconv_hb<2>", "k_edgeconv_hb<1>", "k_mlp1_apply_b"):
            r = by.get(k)
            if r is None:
                problems.append(f"solo CSV: no row for {k}")
                continue
            if r["min_us"] < (1 - band) * r["avg_us"] or r["max_us"] > (1 + band) * r["avg_us"]:
                problems.append(f"solo CSV: {k} min {r['min_us']:.1f} / avg {r['avg_us']:.1f} / max {r['max_us']:.1f} us over {r['calls']} calls: "
                                f"not one launch size (band {band:.0%})")
    return (not problems), problems


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


def kernel_table(tag, pmc, solo_rows, B):
    """DESIGN.md section 4's per-kernel table from the files (markdown)."""
    v = (pmc or {}).get("valu_insts_per_scene_launch", {})
    hb = (pmc or {}).get("hbm_bytes_per_scene_launch", {})
    hu = (pmc or {}).get("hbm_bytes_uncorrected_per_scene_launch", {})
    mb = (pmc or {}).get("mfma_busy_share", {})
    wv = (pmc or {}).get("mean_waves_per_simd", {})
    iss = (pmc or {}).get("issue_share", {})
    out = [f"<!-- generated by tools/summarise_profiles.py {tag} from profiles/{tag}_solo_batched_kernel_stats.csv and profiles/{tag}_pmc_kernels.json -->",
           f"| kernel (batched launch of {B} scenes, one group alone on the GPU) | launches | µs per launch avg (min – max) | **µs per scene** | VALU M / scene | HBM MB / scene (2F+W; as reported) | MFMA busy | waves / SIMD | issuing / `s_waitcnt` |",
           "|---|---|---|---|---|---|---|---|---|"]
    tot = 0.0
    per_scene_calls = {}
    ref_calls = None
    for r in solo_rows:
        if r["kernel"] == "k_mlp1_apply_b":
            ref_calls = r["calls"]
    for r in solo_rows:
        k = r["kernel"]
        if not (k.endswith("_b") or "_b<" in k or "_hb<" in k):
            continue
        a = ALIAS.get(k, k)
        mult = (r["calls"] / ref_calls) if ref_calls else 1.0       # launches per group-step (k_edge_distance_b: 3, k_layer_layout_b: 2 ...)
        us_scene = r["avg_us"] / B * mult
        tot += us_scene
        per_scene_calls[k] = mult
        f = lambda x, s=1.0, fmt="%.1f": (fmt % (x * s)) if x is not None else ""
        i = iss.get(a, {})
        out.append("| `%s`%s | %d | %.1f (%.1f – %.1f) | **%.1f** | %s | %s | %s | %s | %s |" % (
            a, (" x%g" % mult) if abs(mult - 1) > 1e-9 else "", r["calls"], r["avg_us"], r["min_us"], r["max_us"], us_scene,
            f(v.get(a), 1e-6 * mult), (f(hb.get(a), 1e-6 * mult) + "; " + f(hu.get(a), 1e-6 * mult)) if a in hb else "",
            f(mb.get(a), 100, "%.1f %%") if a in mb and mb.get(a, 0) > 0.001 else "", f(wv.get(a), 1, "%.2f"),
            ("%.0f %% / %.0f %%" % (100 * i.get("issuing", 0), 100 * i.get("waiting_s_waitcnt", 0))) if i else ""))
    out.append(f"| **sum of the batched kernels** | | | **{tot:.1f}** | | | | | |")
    return "\n".join(out) + "\n", tot


def main(tag):
    os.makedirs(P, exist_ok=True)
    bench = stats("prof_bench", f"{tag}_bench_kernel_stats.csv", "python bench.py --no-cpu-baseline --no-files --parity-scenes 0 (default engine shape), 150k/1.5k scenes")
    solo = stats("prof_solo8", f"{tag}_solo_batched_kernel_stats.csv", "python bench.py --groups 1 --per-group 8 --no-cpu-baseline --no-files --parity-scenes 0: one group, 8 scenes per batched launch, nothing else on the GPU, no parity legs in the process")
    train = stats("prof_train", f"{tag}_train_kernel_stats.csv", "python tools/time_train.py --steps 6: the training step (forward with tape + loss + backward + SGD), 150k/1.5k scenes, one scene per step")

    passes = {}
    for name in ("sq", "mfma", "fetch", "write"):
        p = os.path.join(G, f"{tag}_pmc_{name}.json")
        if os.path.exists(p):
            passes[name] = json.load(open(p))
    out = None
    if passes:
        eng = next(iter(passes.values()))["engine"]
        B = int(eng.split(" x ")[1].split()[0])
        out = {"configuration": f"solo batched: {eng}, python bench.py --steps 2 --warmup 1 --parity-scenes 0 under rocprofv3 --pmc (one counter set per pass, --kernel-trace only)",
               "scenes_per_launch": B, "points": 150000, "segments": 1500,
               "note": "per SCENE-launch = per batched launch / scenes per launch.  hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024: on gfx950 FETCH_SIZE reports half "
                       "the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); hbm_bytes_uncorrected uses FETCH_SIZE as reported.  MFMA busy share = "
                       "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); mean waves per SIMD = 4 x SQ_WAVE_CYCLES / 1024 / kernel cycles.  "
                       "<counter>__min / __max = the smallest / largest per-launch value the pass saw (all launches of a clean collection are full launches).",
               "valu_insts_per_scene_launch": {}, "hbm_bytes_per_scene_launch": {}, "hbm_bytes_uncorrected_per_scene_launch": {}, "mfma_busy_share": {},
               "mean_waves_per_simd": {}, "issue_share": {}, "per_kernel_raw": {}}
        sq, mf = passes.get("sq", {}).get("kernels", {}), passes.get("mfma", {}).get("kernels", {})
        fe, wr = passes.get("fetch", {}).get("kernels", {}), passes.get("write", {}).get("kernels", {})
        lines = []
        for k in sorted(set(sq) | set(mf) | set(fe) | set(wr)):
            if not k.endswith("_b") and "_b<" not in k and "_hb<" not in k:
                continue                                           # batched kernels only
            name = ALIAS.get(k, k)
            raw = {}
            for src in (sq, mf, fe, wr):
                raw.update({n: v for n, v in src.get(k, {}).items() if n != "launches"})
            launches = max((src.get(k, {}).get("launches", 0) for src in (sq, mf, fe, wr)), default=0)
            out["per_kernel_raw"][name] = {"launches": launches, **{n: round(v, 1) for n, v in raw.items()}}
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
        with open(os.path.join(P, f"{tag}_sq_counters.txt"), "w") as o:
            o.write(f"# {out['configuration']}\n# SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY, averages per batched launch\n")
            o.write("\n".join(lines) + "\n")
        print("pmc: %d kernels" % len(out["per_kernel_raw"]))

    solo_path = os.path.join(P, f"{tag}_solo_batched_kernel_stats.csv")
    solo_rows = read_stats_csv(solo_path) if os.path.exists(solo_path) else None
    ok, problems = check_summary(out, solo_rows)
    if out is not None:
        out["checks"] = {"ok": ok, "problems": problems,
                         "what": "every launch the summary averages is a full launch of scenes_per_launch scenes (tools/summarise_profiles.py: check_summary)"}
        json.dump(out, open(os.path.join(P, f"{tag}_pmc_kernels.json"), "w"), indent=1, sort_keys=True)
    if solo_rows:
        table, tot = kernel_table(tag, out, solo_rows, (out or {}).get("scenes_per_launch", 8))
        open(os.path.join(P, f"{tag}_kernel_table.md"), "w").write(table)
        print("kernel table: sum of the batched kernels %.1f us per scene" % tot)
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
    if not ok:
        print("summarise_profiles: THE SUMMARY DOES NOT HOLD FULL LAUNCHES ONLY:", file=sys.stderr)
        for p_ in problems:
            print("   " + p_, file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "r06"))
