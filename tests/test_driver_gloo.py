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


# ---- bench.py: its own reduction and its launcher ---------------------------------------------------------------
def _bench_reduce_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo")
    mine = list(range(rank, 1201, world))                        # bench.py --scenes-total 1201: scene i -> rank i mod W
    vec = np.zeros(165)
    for i in mine:
        s, n_, a = _stub_forward(i)
        vec[:80] += s.reshape(-1); vec[80:160] += n_.reshape(-1); vec[160:164] += a; vec[164] += 1
    out = bench.reduce_accumulators(vec, world, "gloo")
    comm = bench.comm_probe(world, "gloo", None)                  # the bench line's `comm` object: what the collective layer saw
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put((out.tolist(), comm))


def test_bench_reduction_two_ranks_equal_one_rank():
    """bench.py's end-of-run all-reduce of the 165 float64 accumulators over a 1201-scene shard (configs[3]) at world
    size 2 (gloo) equals the single-process sums, and every scene is counted exactly once."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, comm = q.get(timeout=180)
    got = np.asarray(got)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert comm["backend"] == "gloo" and comm["world_size"] == 2 and comm["allreduce_of_ones"] == 2.0 and "error" not in comm, comm
    want = np.zeros(165)
    for i in range(1201):
        s, n_, a = _stub_forward(i)
        want[:80] += s.reshape(-1); want[80:160] += n_.reshape(-1); want[160:164] += a; want[164] += 1
    assert got[164] == 1201 and np.array_equal(got[:160], want[:160]) and np.allclose(got[160:164], want[160:164], rtol=0, atol=1e-9)


def test_comm_probe_without_a_process_group_makes_its_own_one_rank_communicator():
    """bench.py at N = 1 without a launcher: the probe builds a one-rank communicator of the asked backend, reduces over it and removes it."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    assert not dist.is_initialized()
    comm = bench.comm_probe(1, "gloo", None)
    assert comm["world_size"] == 1 and comm["allreduce_of_ones"] == 1.0 and comm["expected"] == 1 and "error" not in comm, comm
    assert not dist.is_initialized()


def test_bench_refuses_a_multi_gpu_label_without_the_gpus():
    """`python bench.py --gpus 2` on a box with fewer than 2 HIP devices must fail loudly (round-1 bug: it measured one GPU
    and printed n_gpus: 1); and a rank whose WORLD_SIZE disagrees with --gpus must refuse as well."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with < 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "only" in r.stderr and '{"metric"' not in r.stdout
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and '{"metric"' not in r.stdout


# ---- run_infer.log against the reference's own transcript (tools/capture_transcript.py) ---------------------------------
def _transcript_spec(golden_index):
    meta = __import__("json").load(open(os.path.join(ROOT, "tests", "golden", "transcript.json")))
    return meta


def _log_from_first_progress_line(path):
    text = open(path).read()
    at = text.index("Infer(0001/")
    return text[:at], text[at:]


@pytest.mark.parametrize("mode", ["ins_infer", "sem_infer"])
def test_driver_log_equals_the_reference_transcript_on_cpu(tmp_path, golden_index, weight_sets, mode):
    """`tests/golden/transcript_<mode>.log` is what the reference's OWN `infer()` (infer.py:127-190) logged over an eight-scene tree (the four full
    fixtures among four small scenes), world size 1, its DistributedSampler's shuffled order.  The driver's loop, accumulation and format strings
    (`seggroup_amd/infer.py`: scene_indices, Accumulator, progress_line, final_report) must reproduce that text BYTE FOR BYTE from per-scene metric
    tensors -- here the reference's captured ones (fixtures) and the oracle's (small scenes); the GPU twin of this test feeds the HIP path's."""
    from oracle import cpu_ref
    from seggroup_amd import infer, synthetic
    from conftest import load_golden
    meta = _transcript_spec(golden_index)
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "dataset", "scannet"))
    with open(os.path.join(root, "dataset", "scannet", "scannetv2_train.txt"), "w") as f:
        f.write("".join(s["name"] + "\n" for s in meta["scenes"]))
    fixture_of = {(e["n"], e["s"], e["seed"]): k for k, e in golden_index.items()}
    pre = "ins" if mode == "ins_infer" else "sem"

    def forward(i):
        s = meta["scenes"][i]
        fx = fixture_of.get((s["n"], s["s"], s["seed"]))
        if fx is not None:
            g = load_golden(fx)
            return g[f"{pre}.metric.0"], g[f"{pre}.metric.1"], g[f"{pre}.metric.2"]
        r = cpu_ref.forward_scene(synthetic.make_scene(s["n"], s["s"], s["seed"], name=s["name"], **s["kw"]), weight_sets[mode], mode)
        return r["metrics"]
    args = infer.build_parser().parse_args(["-n", "exp", f"--{mode}", "--root", root, "--world-size", "1", "--sampler", "reference"])
    assert infer.scene_indices(len(meta["scenes"]), 0, 1, "reference") == meta[mode]["sampler_order"]
    infer.run_worker(0, 1, args, forward_fn=forward)
    _, tail = _log_from_first_progress_line(os.path.join(root, "checkpoints", "exp", "run_infer.log"))
    want = open(os.path.join(ROOT, "tests", "golden", f"transcript_{mode}.log")).read()
    assert tail == want
