#!/usr/bin/env python3
"""Golden vectors for the pre-processing path from the REAL reference (build container only).

Imports the unmodified `seggroup/dataset/scannet/util.py` from /root/reference (chainer / plyfile stubbed: the build
image has neither), feeds it synthetic raw scans (`seggroup_amd.synthetic.make_raw_scan`) through a stand-in for the
`PlyData` object and a `segs.json` file, and runs, in a scratch directory,
    generate_pointcloud_pth            -> .pcl.pth / .map.pth / .unmap.pth   (util.py:633-693; get_unmapper 538-550)
    generate_seg_labels_and_ds_set     -> .seg.txt / .seg.json               (util.py:174-220)
    generate_mesh_adjcency_pth         -> adj/mesh/{raw,resampled} .adj.pth  (util.py:771-811)
`torch.randperm` is replaced for the duration of the call by the scan's own permutation, so the sampling is
reproducible on the GPU box.  Only inputs' seeds and OUTPUTS are stored (tests/golden/prep_*.npz, digests for the
large case); nothing from /root/reference is copied and this script never runs on the GPU box.

usage: python tools/capture_prepare.py [--only NAME]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference/seggroup/dataset/scannet"

# name -> lattice, seed, num_points, whether full outputs are stored
FIXTURES = {
    "prep_sub_3k": dict(w=60, h=50, seed=7, num_points=2000, full=True),        # V > num_points: unsampled vertices, get_unmapper
    "prep_rep_1k": dict(w=40, h=30, seed=8, num_points=3000, full=True),        # V < num_points: 2 full copies + remainder
    "prep_exact_2k": dict(w=50, h=40, seed=9, num_points=2040, full=True),      # num_points == V (dup_frac 2 %): identity mapper
    "prep_sub_60k": dict(w=300, h=200, seed=10, num_points=40000, full=False),  # digests only
}


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load_reference_util():
    chainer = types.ModuleType("chainer")
    chainer.cuda = types.ModuleType("chainer.cuda")
    sys.modules["chainer"], sys.modules["chainer.cuda"] = chainer, chainer.cuda
    ply = types.ModuleType("plyfile")
    ply.PlyData = type("PlyData", (), {})
    sys.modules["plyfile"] = ply
    sys.path.insert(0, REF)
    import util                                    # the reference module, unmodified
    return util


class _Element:
    def __init__(self, cols):
        self._c = cols
        self.count = len(next(iter(cols.values())))

    def __getitem__(self, k):
        return self._c[k]


def fake_plydata(scan):
    return {"vertex": _Element({"x": scan.xyz[:, 0], "y": scan.xyz[:, 1], "z": scan.xyz[:, 2],
                                "red": scan.rgb[:, 0], "green": scan.rgb[:, 1], "blue": scan.rgb[:, 2]}),
            "face": _Element({"vertex_indices": [f for f in scan.faces]})}


def capture(util, scan, num_points):
    import torch
    ply = fake_plydata(scan)
    out = {}
    with tempfile.TemporaryDirectory(prefix="sgprep_") as td:
        scene_path = os.path.join(td, "scans", scan.name)
        os.makedirs(scene_path)
        with open(os.path.join(scene_path, scan.name + "_vh_clean_2.0.010000.segs.json"), "w") as f:
            json.dump({"segIndices": scan.seg_indices.tolist()}, f)
        cwd = os.getcwd()
        os.chdir(td)
        real_randperm = torch.randperm
        try:
            torch.randperm = lambda n, *a, **k: torch.from_numpy(scan.perm[:n].copy())
            util.generate_pointcloud_pth(scene_path, 5, num_points, ply)
        finally:
            torch.randperm = real_randperm
        try:
            util.generate_seg_labels_and_ds_set(scene_path)
            util.generate_mesh_adjcency_pth(scan.name, ply)
            d = os.path.join("data", "resampled", scan.name)
            out["pcl"] = torch.load(os.path.join(d, scan.name + ".pcl.pth")).numpy()
            out["map"] = torch.load(os.path.join(d, scan.name + ".map.pth")).numpy()
            out["unmap"] = torch.load(os.path.join(d, scan.name + ".unmap.pth")).numpy()
            out["info"] = torch.load(os.path.join(d, scan.name + ".info.pth")).numpy()
            out["adj_raw"] = torch.load(os.path.join("adj", "mesh", "raw", scan.name, scan.name + ".adj.pth")).numpy()
            out["adj_resampled"] = torch.load(os.path.join("adj", "mesh", "resampled", scan.name, scan.name + ".adj.pth")).numpy()
            out["seg_txt"] = open(os.path.join("label", "real", "raw", scan.name, scan.name + ".seg.txt"), "rb").read()
            out["seg_json"] = open(os.path.join("label", "real", "resampled", scan.name, scan.name + ".seg.json"), "rb").read()
        finally:
            os.chdir(cwd)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    from oracle import prep_ref
    from seggroup_amd import synthetic
    util = _load_reference_util()
    idx_path = os.path.join(a.out, "prep_index.json")
    index = json.load(open(idx_path)) if os.path.exists(idx_path) else {}
    for name, fx in FIXTURES.items():
        if a.only and a.only != name:
            continue
        scan = synthetic.make_raw_scan(fx["w"], fx["h"], fx["seed"], name=name)
        t = time.time()
        got = capture(util, scan, fx["num_points"])
        # the restatement against the reference, on the spot
        pcl, mapper, unmap = prep_ref.sample_points(scan.xyz, scan.rgb, fx["num_points"], scan.perm)
        missing = np.setdiff1d(np.arange(scan.xyz.shape[0]), mapper)
        ties = prep_ref.tie_rows(scan.xyz[missing], pcl[:, :3]) if missing.size else np.zeros(0, bool)
        adj_raw, adj_res = prep_ref.get_adj_from_mesh(scan.faces, got["unmap"])
        raw_lab, lists = prep_ref.segment_lists(scan.seg_indices, got["map"])
        checks = {
            "pcl": bool(np.array_equal(pcl, got["pcl"])), "map": bool(np.array_equal(mapper, got["map"])),
            "unmap_outside_ties": bool(np.array_equal(np.delete(unmap, missing[ties]), np.delete(got["unmap"], missing[ties]))),
            "unmap_everywhere": bool(np.array_equal(unmap, got["unmap"])),
            "adj_raw": bool(np.array_equal(adj_raw, got["adj_raw"])), "adj_resampled": bool(np.array_equal(adj_res, got["adj_resampled"])),
            "seg_txt": "".join("%d\n" % v for v in raw_lab).encode() == got["seg_txt"],
            "seg_json": prep_ref.seg_json_text(lists).encode() == got["seg_json"],
        }
        print(f"[{name}] V={scan.xyz.shape[0]} F={scan.faces.shape[0]} N={fx['num_points']} unsampled={missing.size} "
              f"tie rows={int(ties.sum())}  reference {time.time() - t:.1f}s  oracle == reference: {checks}")
        entry = {"w": fx["w"], "h": fx["h"], "seed": fx["seed"], "num_points": fx["num_points"], "V": int(scan.xyz.shape[0]),
                 "F": int(scan.faces.shape[0]), "unsampled": int(missing.size), "tie_rows": missing[ties].tolist(),
                 "oracle_equals_reference": checks,
                 "sha": {k: (hashlib.sha256(v).hexdigest() if isinstance(v, bytes) else sha(v)) for k, v in got.items()},
                 "shapes": {k: (len(v) if isinstance(v, bytes) else list(v.shape)) for k, v in got.items()}}
        if name == "prep_sub_3k":
            # get_adj_from_pointcloud (util.py:814-834; nothing in the reference calls it): run on the sampled cloud, k = 10
            import torch
            pc = util.get_adj_from_pointcloud(torch.from_numpy(got["pcl"][:, :3].copy()), k=10).numpy()
            mine, tie = prep_ref.get_adj_from_pointcloud(got["pcl"], k=10)
            keep = lambda e: e[~(tie[e[:, 0]] | tie[e[:, 1]])]
            checks["pointcloud_adj_outside_ties"] = bool(np.array_equal(keep(mine), keep(pc)))
            checks["pointcloud_adj_everywhere"] = bool(np.array_equal(mine, pc))
            print(f"[{name}] get_adj_from_pointcloud: {pc.shape[0]} rows, {int(tie.sum())} points with tied scores, oracle == reference: "
                  f"{checks['pointcloud_adj_outside_ties']} (everywhere: {checks['pointcloud_adj_everywhere']})")
            np.savez_compressed(os.path.join(a.out, "prep_pointcloud_adj.npz"), adj=pc.astype(np.int32), tie_points=np.nonzero(tie)[0].astype(np.int32))
            entry["oracle_equals_reference"] = checks
        if fx["full"]:
            np.savez_compressed(os.path.join(a.out, name + ".npz"), pcl=got["pcl"], map=got["map"], unmap=got["unmap"], info=got["info"],
                                adj_raw=got["adj_raw"], adj_resampled=got["adj_resampled"],
                                seg_txt=np.frombuffer(got["seg_txt"], dtype=np.uint8), seg_json=np.frombuffer(got["seg_json"], dtype=np.uint8))
        index[name] = entry
    with open(idx_path, "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
