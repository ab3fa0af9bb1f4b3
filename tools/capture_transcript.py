#!/usr/bin/env python3
"""`run_infer.log` as the REAL reference writes it (build container only): VERDICT round 5, item 9 / SURVEY.md 8c.

Imports the unmodified reference `seggroup/infer.py` (+ its `model.py`, `data.py`, `util.py`) from /root/reference with the harness-side shims of
tools/capture_reference.py (chainer / plyfile stubs, a torch proxy whose `.device('cuda')` is the CPU device) plus two of the same kind for the
driver -- `Tensor.cuda()` returns the tensor itself on this CPU-only box, and the `DistributedDataParallel` wrapper of `main_worker` (which needs
a GPU) is a two-line object with `.module` and `__call__` -- and calls the reference's OWN `infer()` (infer.py:127-190) with what its
`main_worker` (infer.py:79-124) builds for it: the reference's `ScanNet` dataset over a tree in the reference's on-disk formats, a real
`DistributedSampler` (one replica: shuffled, seed 0, epoch 0) behind a `DataLoader`, the reference's `IOStream`, a one-rank gloo process group
for the three `dist.all_reduce` calls.  What `infer()` logs -- one `Infer(i/n)` line per scene in the sampler's order, the `==> Infer` line,
the per-class tables of `print_class_iou` (infer.py:63-76) -- is stored verbatim:

    tests/golden/transcript_<mode>.log     the text
    tests/golden/transcript.json           the scenes of the tree (generator arguments), the order the sampler gave, `Network parameters`

tests/test_gpu_scene.py::test_run_infer_log_equals_the_reference_transcript writes the same tree on the GPU box, runs `seggroup_amd.infer`
with `--sampler reference` and compares its `run_infer.log` from the first `Infer(` line on, byte for byte.  Nothing of the reference is copied:
the fixture is its OUTPUT on committed generator arguments.

usage: python tools/capture_transcript.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

# the tree: the four full fixtures of tests/golden/index.json among four ragged small scenes
EXTRA = [dict(n=3000 + 379 * i, s=30 + 4 * i, seed=81000 + i, kw=({"dup_frac": 0.05} if i % 2 == 0 else {})) for i in range(1, 5)]
FIXTURES = ["tiny_4k", "small_20k", "tiny_dup_4k", "island_20k"]


def transcript_scenes(index):
    """[(name, n, s, seed, kw)] -- the test rebuilds the same list"""
    out = []
    for i in range(8):
        e = index[FIXTURES[i // 2]] if i % 2 == 0 else EXTRA[i // 2]
        out.append((f"scene{i:04d}_00", e["n"], e["s"], e["seed"], e["kw"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    import capture_reference as cr
    torch_proxy = cr._install_shims()
    np.seterr(divide="ignore", invalid="ignore")                 # reference infer.py:28
    import torch
    import torch.distributed as dist
    torch.Tensor.cuda = lambda self, *a, **k: self               # harness shim: no GPU in the build container
    import model as model_mod                                    # the reference's seggroup/model.py
    model_mod.torch = torch_proxy
    import infer as ref_infer                                    # the reference's seggroup/infer.py (its __main__ block does not run)
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from seggroup_amd import synthetic, weights as W

    index = json.load(open(os.path.join(args.out, "index.json")))
    spec = transcript_scenes(index)
    scenes = [synthetic.make_scene(n, s, seed, name=name, **kw) for name, n, s, seed, kw in spec]
    wsets = {"ins_infer": W.load_npz(os.path.join(args.out, "weights_g2.npz")), "sem_infer": W.load_npz(os.path.join(args.out, "weights_g1.npz"))}
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:23459", world_size=1, rank=0)
    meta = {"scenes": [dict(name=name, n=n, s=s, seed=seed, kw=kw) for name, n, s, seed, kw in spec],
            "what": "output of the reference's own infer() (infer.py:127-190) over this tree, world size 1, DistributedSampler(shuffle=True, seed 0, epoch 0); "
                    "tools/capture_transcript.py"}
    cwd = os.getcwd()
    for mode in ("ins_infer", "sem_infer"):
        with tempfile.TemporaryDirectory() as wd:
            synthetic.write_reference_tree(wd, scenes)
            os.chdir(wd)
            try:
                for kind in ("segment", "instance", "semantic"):     # SURVEY 8c item 5 (the default of 150000 only matters beyond it)
                    getattr(model_mod, f"export_{kind}_label").__defaults__ = (150000,)
                torch.manual_seed(1)
                net = model_mod.SegModel(exp_name="cap", cuda=False, sem_infer=(mode == "sem_infer"), ins_infer=(mode == "ins_infer"))
                missing = net.load_state_dict(W.to_state_dict(wsets[mode], prefix=""), strict=False)
                assert not [k for k in missing.missing_keys if "classifier" not in k and "running" not in k and "num_batches" not in k], missing
                n_params = sum(x.nelement() for x in net.parameters())

                class DDPLike:                                       # main_worker wraps the model in DistributedDataParallel (GPU only)
                    def __init__(self, m):
                        self.module = m

                    def __call__(self, *a):
                        return self.module(*a)

                ds = ref_infer.ScanNet(label_style="manual")
                sampler = DistributedSampler(ds, num_replicas=1, rank=0)             # infer.py:98
                loader = DataLoader(ds, num_workers=0, batch_size=1, shuffle=False, pin_memory=False, sampler=sampler)
                os.makedirs(os.path.join("checkpoints", "cap"), exist_ok=True)
                io = ref_infer.IOStream(os.path.join("checkpoints", "cap", "run_infer.log"))
                a = argparse.Namespace(sem_infer=(mode == "sem_infer"), ins_infer=(mode == "ins_infer"), rank=0, gpu=0, exp_name="cap")
                t0 = time.time()
                ref_infer.infer(loader, sampler, DDPLike(net), 0, 1, a, io)
                io.close()
                sampler.set_epoch(0)
                order = list(iter(sampler))
                text = open(os.path.join("checkpoints", "cap", "run_infer.log")).read()
            finally:
                os.chdir(cwd)
        open(os.path.join(args.out, f"transcript_{mode}.log"), "w").write(text)
        meta[mode] = {"network_parameters": int(n_params), "sampler_order": [int(i) for i in order], "lines": text.count("\n"),
                      "reference_seconds": round(time.time() - t0, 1)}
        print(f"[{mode}] {text.count(chr(10))} lines, {time.time() - t0:.1f} s; sampler order {order}")
        print(text[:400])
    dist.destroy_process_group()
    json.dump(meta, open(os.path.join(args.out, "transcript.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
