#!/usr/bin/env python3
"""Pack-build rate (scenes/s) of the native builder and of the Python builder by thread count, on a tmpfs tree of 150k-point scenes (no GPU involved).

    python3 tools/time_pack_build.py [--scenes 512] [--threads 1,8,16,32,64,128] [--base /dev/shm]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SEGGROUP_HOST_ONLY", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=512)
    ap.add_argument("--threads", default="1,8,16,32,64,128")
    ap.add_argument("--base", default="/dev/shm")
    ap.add_argument("--distinct", type=int, default=32)
    a = ap.parse_args()
    from seggroup_amd import cache, synthetic
    root = tempfile.mkdtemp(prefix="sg_packs_", dir=a.base)
    try:
        base = [synthetic.make_scene(150000, 1500, 20004 + i, name=f"scene{i:04d}_00") for i in range(4)]
        scenes = [synthetic.Scene(f"scene{i:04d}_00", b.data, b.weak_label, b.seg, b.adj, b.unmap, b.gt) for i in range(a.scenes) for b in [base[i % 4]]]
        d_ = min(a.distinct, len(scenes))
        synthetic.write_reference_tree(root, scenes[:d_])
        base_ = os.path.join(root, "dataset", "scannet")
        kinds = [(("data", "resampled"), (".pcl.pth", ".info.pth", ".unmap.pth")), (("label", "seg", "manual", "resampled"), (".label.pth",)),
                 (("label", "real", "resampled"), (".seg.json",)), (("label", "real", "raw"), (".label.pth",)), (("adj", "mesh", "resampled"), (".adj.pth",))]
        for i in range(d_, len(scenes)):
            src, dst = scenes[i % d_].name, scenes[i].name
            for sub, exts in kinds:
                os.makedirs(os.path.join(base_, *sub, dst), exist_ok=True)
                for e in exts:
                    if os.path.exists(os.path.join(base_, *sub, src, src + e)):
                        os.symlink(os.path.join(base_, *sub, src, src + e), os.path.join(base_, *sub, dst, dst + e))
        names = [s.name for s in scenes]
        for mode in ("native", "python"):
            os.environ["SG_PACK_BUILD"] = mode
            for t in [int(x) for x in a.threads.split(",")]:
                if mode == "python" and t not in (8, 32):
                    continue
                shutil.rmtree(os.path.join(base_, "cache"), ignore_errors=True)
                t0 = time.time()
                n = cache.build_missing(root, names, workers=t)
                dt = time.time() - t0
                print(f"{mode:7s} {t:4d} threads: {n / dt:8.1f} packs/s ({n} packs, {dt:.2f} s)", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
