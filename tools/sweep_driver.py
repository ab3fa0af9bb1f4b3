#!/usr/bin/env python3
"""Warm `seggroup_amd.infer` runs over one tmpfs tree under a list of configurations (environment knobs of the loader / engine, driver flags):
overall and steady scenes/s of each, twice.  For finding what the driver's steady rate is bound by.

    python3 tools/sweep_driver.py [--scenes 2048] [--format npy] CONFIG [CONFIG ...]
    CONFIG = name[,ENV=VALUE ...][,--flag=value ...]       e.g.  base  copies1,SG_LOADER_COPIES=1  inflight96,--inflight=96
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=2048)
    ap.add_argument("--format", default="npy")
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--pre-bind", type=int, default=-1, help="NUMA node whose CPUs this process is confined to BEFORE it writes the tree and the packs "
                                                              "(tmpfs pages are placed by first touch: are the packs on the GPU's node or on the other one?)")
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    if a.pre_bind >= 0:
        from seggroup_amd.numa import parse_cpulist
        os.sched_setaffinity(0, parse_cpulist(open(f"/sys/devices/system/node/node{a.pre_bind}/cpulist").read()))
        print(f"pre-bound to node {a.pre_bind}: {len(os.sched_getaffinity(0))} CPUs", flush=True)
    import torch
    from seggroup_amd import infer, synthetic, weights
    root = tempfile.mkdtemp(prefix="sg_sweep_", dir="/dev/shm")
    try:
        base = [synthetic.make_scene(150000, 1000, 20004 + i, name=f"scene{i:04d}_00") for i in range(4)]
        scenes = [synthetic.Scene(f"scene{i:04d}_00", b.data, b.weak_label, b.seg, b.adj, b.unmap, b.gt) for i in range(a.scenes) for b in [base[i % 4]]]
        d_ = 32
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
        with open(os.path.join(base_, "scannetv2_train.txt"), "w") as f:
            f.write("".join(s.name + "\n" for s in scenes))
        ck = os.path.join(root, "checkpoints", "exp", "models")
        os.makedirs(ck)
        torch.save({"state_dict": weights.to_full_state_dict(weights.make_weights(1, bn1_gamma=2.0))}, os.path.join(ck, "last.t7"))
        common = ["-n", "exp", "--ins_infer", "--root", root, "--world-size", "1", "--out-format", a.format, "-j", str(a.workers)]

        def run(extra):
            shutil.rmtree(os.path.join(root, "results"), ignore_errors=True)
            args = infer.build_parser().parse_args(common + extra)
            t = time.time()
            r = infer.run_worker(0, 1, args)
            dt = time.time() - t
            steady = (a.scenes - r["first_batch"]) / max(r["elapsed_s"] - r["startup_s"], 1e-9) if r.get("startup_s") is not None else 0.0
            return a.scenes / dt, steady, r.get("startup_s", 0.0)
        run([])                                                  # cold: packs, HIP context
        run([])
        for rep in range(2):
            for cfg in a.configs:
                parts = cfg.split(",")
                env = {p.split("=", 1)[0]: p.split("=", 1)[1] for p in parts[1:] if not p.startswith("--")}
                flags = [x for p in parts[1:] if p.startswith("--") for x in p.split("=", 1)]
                saved = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    o, s, su = run(flags)
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                print(f"{parts[0]:18s} overall {o:7.1f}  steady {s:7.1f}  start-up {su:.3f} s", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
