"""BASELINE.json configs[3] on the hardware a test box has: the scene-parallel driver (`seggroup_amd.infer`, infer.py:79-124,
149-176 in the reference) over one scene tree at world size 1 and at world size 2 -- two processes with a gloo rendezvous on
127.0.0.1, both ranks on cuda:0, each with its own scene engine, shard and writer pool.  Sharding must change nothing: every
scene's label files are byte-identical between the two runs, and rank 0's all-reduced metric vector equals the one-process
summary, for the `i mod W` sharding and for the reference's DistributedSampler order."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, make_fixture_scene

pytestmark = pytest.mark.gpu


def _rank_worker(rank, world, root, port, sampler, exp, q):
    sys.path.insert(0, ROOT)
    from seggroup_amd import infer
    args = infer.build_parser().parse_args(["-n", exp, "--ins_infer", "--root", root, "--backend", "gloo", "--port", str(port), "--sampler", sampler,
                                            "--batch", "5", "--inflight", "4", "-j", "2"])
    r = infer.run_worker(rank, world, args)
    if rank == 0:
        q.put({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in r.items()})


def _tree_digest(root, exp, names):
    out = {}
    for n in names:
        d = os.path.join(root, "results", exp, n, "ins_infer")
        files = sorted(os.listdir(d))
        assert len(files) == 28, (n, files)                                        # 14 vectors x (.txt, .npy)
        out[n] = {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in files}
    return out


@pytest.mark.parametrize("sampler,n_scenes", [("shard", 25), ("reference", 24)])
def test_two_ranks_on_one_gpu_write_the_same_files_as_one_rank(tmp_path, golden_index, weight_sets, sampler, n_scenes):
    import torch
    import torch.multiprocessing as mp
    from seggroup_amd import hip, infer, synthetic, weights
    root = str(tmp_path)
    fixtures = ["tiny_4k", "small_20k", "tiny_dup_4k", "island_20k"]
    scenes = []
    for i in range(n_scenes):                                                      # ragged: 3k-12k points, the four fixtures among them
        if i % 6 == 0 and i // 6 < len(fixtures):
            e = golden_index[fixtures[i // 6]]
            scenes.append(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]))
        else:
            scenes.append(synthetic.make_scene(3000 + 379 * i, 30 + 4 * i, 81000 + i, name=f"scene{i:04d}_00",
                                               **({"dup_frac": 0.05} if i % 5 == 0 else {})))
    synthetic.write_reference_tree(root, scenes)
    names = [s.name for s in scenes]
    for exp in ("w1", "w2"):
        ck = os.path.join(root, "checkpoints", exp, "models")
        os.makedirs(ck)
        torch.save({"state_dict": weights.to_full_state_dict(weight_sets["ins_infer"])}, os.path.join(ck, "last.t7"))
    one = infer.run_worker(0, 1, infer.build_parser().parse_args(
        ["-n", "w1", "--ins_infer", "--root", root, "--world-size", "1", "--sampler", sampler, "--batch", "5", "--inflight", "4", "-j", "2"]))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, root, port, sampler, "w2", q)) for r in range(2)]
    for p in procs:
        p.start()
    two = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # 1. the files: byte-identical per scene, and the fixtures' carry the reference's integers
    a, b = _tree_digest(root, "w1", names), _tree_digest(root, "w2", names)
    bad = [n for n in names if a[n] != b[n]]
    assert not bad, f"scenes whose label files differ between W = 1 and W = 2: {bad}"
    for i, fx in enumerate(fixtures):
        g = load_golden(fx)
        d = os.path.join(root, "results", "w2", names[6 * i], "ins_infer")
        for nm in hip.LABEL_NAMES:
            assert np.array_equal(np.load(os.path.join(d, nm + ".npy")), g[f"ins.label.{nm}"]), (fx, nm)
    # 2. the reduced metric vector: the all-reduce of two shards == the one-process accumulation (integer counts in float64: exact)
    assert two["n"] == one["n"] == n_scenes
    for k in one:
        if k not in ("elapsed_s", "startup_s", "first_batch"):
            assert np.array_equal(np.asarray(one[k]), np.asarray(two[k]), equal_nan=True), k
    log = open(os.path.join(root, "checkpoints", "w2", "run_infer.log")).read()
    assert "==> Infer           Instance mIoU:" in log
