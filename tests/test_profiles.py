"""The committed rocprofv3 summaries must hold FULL launches only (round 5's profiled process also ran the out-of-step
parity leg -- the same batched kernels, one scene per launch -- and every per-launch average of `profiles/r05_*` came out 7.3 % low).
`tools/summarise_profiles.py: check_summary` holds a summary to the launch model; here it runs on the committed files (no GPU)."""
import copy
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarise_profiles as sp  # noqa: E402


def _tag():
    src = open(os.path.join(ROOT, "bench.py")).read()
    return re.search(r'^PROFILE_TAG = "(r\d+)"', src, re.M).group(1)


def _load(tag):
    pj = os.path.join(ROOT, "profiles", f"{tag}_pmc_kernels.json")
    cs = os.path.join(ROOT, "profiles", f"{tag}_solo_batched_kernel_stats.csv")
    return (json.load(open(pj)) if os.path.exists(pj) else None), (sp.read_stats_csv(cs) if os.path.exists(cs) else None)


def test_round5_summaries_are_recognised_as_diluted():
    pmc, solo = _load("r05")
    assert pmc is not None and solo is not None
    ok, problems = sp.check_summary(pmc, solo)
    assert not ok
    text = "\n".join(problems)
    assert "k_mlp1_apply_b: SQ_WAVES = 11,125.0" in text and "12,000" in text           # 33 full launches + 3 one-scene launches
    assert "k_edgeconv<S2X>: SQ_INSTS_MFMA" in text
    assert "solo CSV: k_edgeconv_hb<2>" in text                                          # min 85.8 / avg 419 / max 912 us


def test_the_summaries_bench_py_reads_hold_full_launches_only():
    tag = _tag()
    pmc, solo = _load(tag)
    if pmc is None or solo is None:
        pytest.skip(f"profiles/{tag}_* not collected yet")
    ok, problems = sp.check_summary(pmc, solo)
    assert ok, "\n".join(problems)
    assert pmc["checks"]["ok"] is True
    # the table DESIGN.md section 4 quotes is the generator's output for these files
    table, _ = sp.kernel_table(tag, pmc, solo, pmc["scenes_per_launch"])
    assert table == open(os.path.join(ROOT, "profiles", f"{tag}_kernel_table.md")).read()
    # ... and the bench line committed beside them took its dominant kernel's duration from launches that agree with the CSV
    bj = os.path.join(ROOT, "profiles", f"{tag}_bench.json")
    if os.path.exists(bj):
        line = json.loads(open(bj).read())
        by = {r["kernel"]: r for r in solo}
        csv_ms = by["k_edgeconv_hb<2>"]["avg_us"] / pmc["scenes_per_launch"] * 1e-3
        assert abs(line["roofline"]["ms_per_scene_launch"] - csv_ms) <= 0.03 * csv_ms, (line["roofline"]["ms_per_scene_launch"], csv_ms)
        assert line["roofline"]["traffic"] == pmc["hbm_bytes_per_scene_launch"]["k_edgeconv<S2X>"]
        # ... and its whole-job figures are the PMC file's sums (every batched kernel x its launches per scene) times the line's own rate
        rl, raw = line["roofline"], pmc["per_kernel_raw"]
        base = raw["k_mlp1_apply_b"]["launches"]
        valu = sum(v * raw[k]["launches"] / base for k, v in pmc["valu_insts_per_scene_launch"].items())
        hbm = sum(v * raw[k]["launches"] / base for k, v in pmc["hbm_bytes_per_scene_launch"].items())
        assert abs(rl["whole_job_valu"]["valu_insts_per_scene"] - valu) <= 1 and abs(rl["whole_job_hbm"]["hbm_bytes_per_scene"] - hbm) <= 1
        per_gpu = line["value"] / line["n_gpus"]
        assert abs(rl["whole_job_valu"]["frac"] - valu * per_gpu / 512e9) < 1e-3 and 0.5 < rl["whole_job_valu"]["frac"] < 1.0
        assert abs(rl["whole_job_hbm"]["frac"] - hbm * per_gpu / 8e12) < 1e-3 and rl["whole_job_hbm"]["frac"] < 0.5
        assert rl["frac"] == pytest.approx(rl["achieved"] / rl["peak"], abs=1e-4)


def test_a_summary_with_mixed_launch_sizes_fails():
    tag = _tag()
    pmc, solo = _load(tag)
    if pmc is None or solo is None:                               # build a clean summary by hand from the launch model
        pmc = {"scenes_per_launch": 8, "per_kernel_raw": {"k_mlp1_apply_b": {"SQ_WAVES": 12000.0},
                                                          "k_edgeconv<S2X>": {"SQ_INSTS_MFMA": 8 * 4688 * 570.0, "SQ_WAVES": 2048.0}}}
        solo = [{"kernel": k, "calls": 49, "avg_us": a, "min_us": a * 0.98, "max_us": a * 1.02, "total_ms": 49 * a * 1e-3}
                for k, a in (("k_edgeconv_hb<2>", 730.0), ("k_edgeconv_hb<1>", 260.0), ("k_mlp1_apply_b", 120.0))]
    assert sp.check_summary(pmc, solo)[0]
    bad = copy.deepcopy(pmc)
    bad["per_kernel_raw"]["k_mlp1_apply_b"]["SQ_WAVES"] *= (33 + 3 / 8.0) / 36.0           # 33 full launches and three one-scene launches averaged together
    assert not sp.check_summary(bad, solo)[0]
    bad2 = copy.deepcopy(solo)
    for r in bad2:
        if r["kernel"] == "k_edgeconv_hb<2>":
            r["min_us"] = r["avg_us"] / 8.0
    assert not sp.check_summary(pmc, bad2)[0]
