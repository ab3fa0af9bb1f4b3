#!/usr/bin/env python3
"""End-to-end driver timing on one GPU: `seggroup_amd.infer` over a synthetic reference-layout
tree, (a) per-scene loop reading the reference's files each step (--batch 0), (b) packed fast path, cold (packs are
built) and warm.  Prints one JSON object; files really are written (txt+npy unless --out-format says otherwise).

    python tools/time_driver.py --scenes 16 --points 150000 --out gpurun_out/driver.json
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--segments", type=int, default=1000)
    ap.add_argument("--out-format", default="txt,npy", help="label file formats; several sets separated by ';' share one tree (e.g. 'npy;txt,npy')")
    ap.add_argument("--base", default="/tmp", help="where the input tree and the results live (/dev/shm = tmpfs)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--inflight", type=int, default=0, help="0 = the driver's own default")
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-nopack", action="store_true", help="skip the legs that stage from the reference's files")
    ap.add_argument("--skip-loop", action="store_true", help="skip the per-scene loop legs (7 scenes/s: slow for many scenes)")
    ap.add_argument("--numa", default="auto", choices=["auto", "off"])
    ap.add_argument("--label-transfer", default="tables", choices=["tables", "full"])
    ap.add_argument("--distinct", type=int, default=0, help="write this many scenes' files for real and symlink the others' to them (0 = write "
                                                            "every scene: 2,048 scenes take five minutes of torch.save on the GPU box)")
    a = ap.parse_args()

    import torch
    from seggroup_amd import infer, synthetic, weights

    root = tempfile.mkdtemp(prefix="sg_driver_", dir=a.base)
    try:
        t0 = time.time()
        base = [synthetic.make_scene(a.points, a.segments, 20004 + i, name=f"scene{i:04d}_00") for i in range(min(a.scenes, 4))]
        scenes = []
        for i in range(a.scenes):          # distinct names, 4 distinct geometries (generation is the slow part here)
            b = base[i % len(base)]
            scenes.append(synthetic.Scene(f"scene{i:04d}_00", b.data, b.weak_label, b.seg, b.adj, b.unmap, b.gt))
        d_ = a.distinct if 0 < a.distinct < len(scenes) else len(scenes)
        synthetic.write_reference_tree(root, scenes[:d_])
        if d_ < len(scenes):
            base_ = os.path.join(root, "dataset", "scannet")
            kinds = [(("data", "resampled"), (".pcl.pth", ".info.pth", ".unmap.pth")), (("label", "seg", "manual", "resampled"), (".label.pth",)),
                     (("label", "real", "resampled"), (".seg.json",)), (("label", "real", "raw"), (".label.pth",)), (("adj", "mesh", "resampled"), (".adj.pth",))]
            for i in range(d_, len(scenes)):
                src, dst = scenes[i % d_].name, scenes[i].name
                for sub, exts in kinds:
                    os.makedirs(os.path.join(base_, *sub, dst), exist_ok=True)
                    for e in exts:
                        os.symlink(os.path.join(base_, *sub, src, src + e), os.path.join(base_, *sub, dst, dst + e))
            with open(os.path.join(base_, "scannetv2_train.txt"), "w") as f:
                for sc_ in scenes:
                    f.write(sc_.name + "\n")
        ck = os.path.join(root, "checkpoints", "exp", "models")
        os.makedirs(ck)
        torch.save({"state_dict": weights.to_full_state_dict(weights.make_weights(1, bn1_gamma=2.0))}, os.path.join(ck, "last.t7"))
        gen_s = time.time() - t0
        out = {"scenes": a.scenes, "points": a.points, "base": a.base, "tree_build_s": round(gen_s, 1)}
        for fmt_w in a.out_format.split(";"):
            fmt, _, wk = fmt_w.partition("@")                      # 'npy@4' = this leg with -j 4
            a.workers = int(wk) if wk else a.workers
            common = ["-n", "exp", "--ins_infer", "--root", root, "--world-size", "1", "--out-format", fmt, "-j", str(a.workers), "--numa", a.numa, "--label-transfer", a.label_transfer]

            def run(extra, keep=False):
                if not keep:
                    shutil.rmtree(os.path.join(root, "results"), ignore_errors=True)
                args = infer.build_parser().parse_args(common + extra)
                t = time.time()
                r = infer.run_worker(0, 1, args)
                return time.time() - t, r

            o = {}
            r0 = None
            if not a.skip_loop:
                run(["--batch", "0"])                              # warm-up: HIP context, page cache
                t, r0 = run(["--batch", "0"])
                o["per_scene_loop"] = {"s": round(t, 3), "scenes_per_s": round(a.scenes / t, 2)}
            fast = ["--batch", str(a.batch)] + (["--inflight", str(a.inflight)] if a.inflight else [])
            if not a.skip_nopack:
                run(fast + ["--no-cache"])                         # warm-up of this leg (engine creation, page cache)
                t, rn = run(fast + ["--no-cache"])
                o["reference_files_no_pack"] = {"s": round(t, 3), "scenes_per_s": round(a.scenes / t, 2)}
            t, r1 = run(fast)
            o["packed_cold"] = {"s": round(t, 3), "scenes_per_s": round(a.scenes / t, 2)}
            # ... which, as this process's FIRST run, also pays for the HIP context, the library's code objects and the first engine: the same cold tree once more
            # (packs removed) in the process as it is now -- what the pack build alone adds to a run
            shutil.rmtree(os.path.join(root, "dataset", "scannet", "cache"), ignore_errors=True)
            t, _ = run(fast)
            o["packed_cold_warm_process"] = {"s": round(t, 3), "scenes_per_s": round(a.scenes / t, 2)}
            # the same run over the files of the last one (a re-run of infer.py: the label files exist and are overwritten, no pages to allocate)
            t, rk = run(fast, keep=True)
            o["packed_warm_overwrite"] = {"s": round(t, 3), "scenes_per_s": round(a.scenes / t, 2), "driver_elapsed_s": round(rk.get("elapsed_s", 0.0), 3)}
            t, r2 = run(fast)
            def leg(t_, r_):
                d_ = {"s": round(t_, 3), "scenes_per_s": round(a.scenes / t_, 2), "driver_elapsed_s": round(r_.get("elapsed_s", 0.0), 3)}
                if r_.get("startup_s") is not None and a.scenes > r_["first_batch"]:
                    # one-off part (first batch staged alone + engine slots + pinned label ring) and the rate behind it
                    d_["startup_s"] = round(r_["startup_s"], 3)
                    d_["steady_scenes_per_s"] = round((a.scenes - r_["first_batch"]) / max(r_["elapsed_s"] - r_["startup_s"], 1e-9), 2)
                return d_
            o["packed_warm"] = leg(t, r2)
            t, r3 = run(fast)
            o["packed_warm_2"] = leg(t, r3)
            ref = r0 if r0 is not None else r1
            skip = ("elapsed_s", "startup_s", "first_batch")
            o["summaries_equal"] = all(repr(ref[k]) == repr(r2[k]) for k in ref if k not in skip)
            o["files_written"] = sum(len(f) for _, _, f in os.walk(os.path.join(root, "results")))
            out[fmt_w] = o
        print(json.dumps(out))
        if a.out:
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            with open(a.out, "w") as f:
                json.dump(out, f, indent=1)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
