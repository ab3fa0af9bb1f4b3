"""N>1 driver logic on CPU: world_size-2 gloo processes shard the scene list, run a stub forward, and
the single end-of-run all-reduce gives rank 0 the same totals as a 1-process run."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_forward(i):
    rng = np.random.default_rng(1000 + i)
    iou_sem = rng.integers(0, 50, (1, 2, 40)).astype(np.float32)
    iou_ins = rng.integers(0, 50, (1, 2, 40)).astype(np.float32)
    return iou_sem, iou_ins, rng.uniform(size=4).astype(np.float32)


def _worker(rank, world, root, port, sampler, q):
    sys.path.insert(0, ROOT)
    from seggroup_amd import infer
    args = infer.build_parser().parse_args(["-n", "exp", "--ins_infer", "--root", root, "--backend", "gloo", "--port", str(port),
                                            "--sampler", sampler])
    r = infer.run_worker(rank, world, args, forward_fn=_stub_forward)
    if rank == 0:
        q.put({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in r.items()})


@pytest.mark.parametrize("sampler", ["shard", "reference"])
def test_two_ranks_equal_one_rank(tmp_path, sampler):
    import torch.multiprocessing as mp
    from seggroup_amd import infer
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "dataset", "scannet"))
    n_scenes = 11
    with open(os.path.join(root, "dataset", "scannet", "scannetv2_train.txt"), "w") as f:
        f.write("".join(f"scene{i:04d}_00\n" for i in range(n_scenes)))
    # sharding covers every scene exactly once; the reference sampler pads like DistributedSampler
    shards = [infer.scene_indices(n_scenes, r, 2, "shard") for r in range(2)]
    assert sorted(shards[0] + shards[1]) == list(range(n_scenes))
    refs = [infer.scene_indices(1201, r, 8, "reference") for r in range(8)]
    assert all(len(x) == 151 for x in refs) and refs[0][:5] == [600, 817, 802, 168, 568]     # SURVEY appendix B probe
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, root, port, sampler, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process expectation over the same multiset of scenes
    acc = infer.Accumulator()
    for r in range(2):
        for i in infer.scene_indices(n_scenes, r, 2, sampler):
            acc.add(*_stub_forward(i))
    want = acc.summary()
    assert got["n"] == want["n"] == (n_scenes if sampler == "shard" else 12)
    assert np.allclose(got["iou_sem"], want["iou_sem"], equal_nan=True) and np.allclose(got["iou_ins"], want["iou_ins"], equal_nan=True)
    assert abs(got["acc_sem"] - want["acc_sem"]) < 1e-12
    log = open(os.path.join(root, "checkpoints", "exp", "run_infer.log")).read()
    assert "==> Infer           Instance mIoU:" in log and "Semantic mIoU (20 classes)" in log and "otherfurniture" in log
