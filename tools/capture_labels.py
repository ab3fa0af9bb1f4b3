#!/usr/bin/env python3
"""Golden outputs of the reference's annotation-derived label producers (build container only): runs the UNMODIFIED functions of
/root/reference/seggroup/dataset/scannet/util.py -- generate_real_labels, generate_weak_labels (manual, maxseg, maxseg with
anno_num 2, rand and mainseg_3 under np.random.seed(1)), generate_real_label_pth, generate_weak_label_pth -- on a synthetic scan
with synthetic annotations (seggroup_amd.synthetic.make_annotations) and stores what they wrote in tests/golden/prep_labels.npz.
usage: python tools/capture_labels.py"""
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import capture_prepare as cp  # noqa: E402

STYLES = (("manual", {}), ("maxseg", {}), ("maxseg", {"anno_num": 2}), ("rand", {}), ("mainseg", {"main_num": 3}))
FIXTURE = dict(w=64, h=48, seed=21, num_points=2000, name="prep_lab_3k", cell=8)


def style_dir(style, kw):
    return style + ("_" + str(kw["main_num"]) if style == "mainseg" else "") + ("_a" + str(kw["anno_num"]) if kw.get("anno_num", 1) > 1 else "")


def write_inputs(td, scan, ann):
    scene_path = os.path.join(td, "scans", scan.name)
    os.makedirs(scene_path, exist_ok=True)
    with open(os.path.join(scene_path, scan.name + "_vh_clean_2.0.010000.segs.json"), "w") as f:
        json.dump({"segIndices": scan.seg_indices.tolist()}, f)
    with open(os.path.join(scene_path, scan.name + ".aggregation.json"), "w") as f:
        json.dump(ann["aggregation"], f)
    with open(os.path.join(td, "scannetv2-labels.combined.tsv"), "w") as f:
        f.write(ann["tsv"])
    os.makedirs(os.path.join(td, "manual"), exist_ok=True)
    with open(os.path.join(td, "manual", scan.name + ".json"), "w") as f:
        json.dump(ann["manual"], f)
    return scene_path


def main():
    import torch
    from seggroup_amd import synthetic
    util = cp._load_reference_util()
    scan = synthetic.make_raw_scan(FIXTURE["w"], FIXTURE["h"], FIXTURE["seed"], name=FIXTURE["name"], cell=FIXTURE["cell"])
    ann = synthetic.make_annotations(scan, 11, blocks_per_row=-(-FIXTURE["w"] // FIXTURE["cell"]))
    ply = cp.fake_plydata(scan)
    out = {}
    with tempfile.TemporaryDirectory(prefix="sglab_") as td:
        scene_path = write_inputs(td, scan, ann)
        cwd = os.getcwd()
        os.chdir(td)
        real_randperm = torch.randperm
        try:
            torch.randperm = lambda n, *a, **k: torch.from_numpy(scan.perm[:n].copy())
            util.generate_pointcloud_pth(scene_path, 5, FIXTURE["num_points"], ply)
            torch.randperm = real_randperm
            util.generate_seg_labels_and_ds_set(scene_path)
            util.generate_real_labels(scene_path)
            util.generate_real_label_pth(scene_path)
            raw = os.path.join("label", "real", "raw", scan.name)
            out["real.ins"] = np.loadtxt(os.path.join(raw, scan.name + ".ins.txt"), dtype=np.int64)
            out["real.sem"] = np.loadtxt(os.path.join(raw, scan.name + ".sem.txt"), dtype=np.int64)
            out["real.label_pth"] = torch.load(os.path.join(raw, scan.name + ".label.pth")).numpy()
            for style, kw in STYLES:
                np.random.seed(1)
                ret = util.generate_weak_labels(scene_path, ply, label_style=style, manual_label_path=os.path.join(td, "manual"), **kw)
                d = style_dir(style, kw)
                util.generate_weak_label_pth(scan.name, d)
                out[f"{d}.ret"] = np.array(ret, dtype=np.int64)
                out[f"{d}.ins"] = np.loadtxt(os.path.join("label", "seg", d, "raw", scan.name, scan.name + ".ins.txt"), dtype=np.int64)
                out[f"{d}.sem"] = np.loadtxt(os.path.join("label", "seg", d, "raw", scan.name, scan.name + ".sem.txt"), dtype=np.int64)
                out[f"{d}.label_pth"] = torch.load(os.path.join("label", "seg", d, "resampled", scan.name, scan.name + ".label.pth")).numpy()
                print(d, "returned", ret, "labeled sampled points", int((out[f"{d}.label_pth"][:, 1] >= 0).sum()), flush=True)
        finally:
            torch.randperm = real_randperm
            os.chdir(cwd)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "prep_labels.npz"), **{k: v.astype(np.int32) for k, v in out.items()})
    print("instances", len(ann["aggregation"]["segGroups"]), "unlabeled vertices", int((out["real.ins"] == 0).sum()), "of", out["real.ins"].shape[0])


if __name__ == "__main__":
    main()
