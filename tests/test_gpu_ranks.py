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


def _rccl_worker(root, port, q):
    """world size 1 over RCCL: the process group exists, so the driver's and the bench's reductions go through an RCCL communicator"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from seggroup_amd import infer
    import bench
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    args = infer.build_parser().parse_args(["-n", "rccl", "--ins_infer", "--root", root, "--backend", "nccl", "--batch", "4", "--inflight", "4", "-j", "2"])
    r = infer.run_worker(0, 1, args, init_dist=False)
    vec = np.arange(165, dtype=np.float64) * 0.5
    red = bench.reduce_accumulators(vec, 1, "nccl", dev)
    # a MAX reduction like the bench's timing line, and a barrier: the other two collectives the GPU paths issue
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    maps = open("/proc/self/maps").read()
    dist.destroy_process_group()
    q.put({"summary": {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in r.items()}, "reduced_equal": bool(np.array_equal(red, vec)),
           "max": float(t.item()), "rccl_mapped": ("librccl" in maps) or ("libnccl" in maps)})


def test_rccl_communicator_at_world_size_one(tmp_path, golden_index, weight_sets):
    """The reference runs one process per GPU over NCCL (infer.py:84-85,234-237).  A test box has one GPU: a world-size-1 `nccl` process
    group still loads RCCL, builds a communicator and runs the path's collectives on the device -- the driver's end-of-run all-reduce
    of the 165 float64 accumulators (infer.run_worker), the bench's reduction, a MAX all-reduce and a barrier -- and must change nothing."""
    import torch
    import torch.multiprocessing as mp
    from seggroup_amd import infer, synthetic, weights
    root = str(tmp_path)
    scenes = [synthetic.make_scene(4000 + 500 * i, 40 + 5 * i, 83000 + i, name=f"scene{i:04d}_00") for i in range(6)]
    synthetic.write_reference_tree(root, scenes)
    for exp in ("plain", "rccl"):
        ck = os.path.join(root, "checkpoints", exp, "models")
        os.makedirs(ck)
        torch.save({"state_dict": weights.to_full_state_dict(weight_sets["ins_infer"])}, os.path.join(ck, "last.t7"))
    one = infer.run_worker(0, 1, infer.build_parser().parse_args(["-n", "plain", "--ins_infer", "--root", root, "--world-size", "1", "--batch", "4",
                                                                  "--inflight", "4", "-j", "2"]))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(root, port, q))
    p.start()
    got = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert got["rccl_mapped"], "the nccl backend did not load RCCL"
    assert got["reduced_equal"] and got["max"] == 3.25
    for k in ("iou_sem", "iou_ins", "acc_sem", "acc_ins", "acc_sem_sel", "acc_ins_sel", "n"):
        assert np.array_equal(np.asarray(one[k], dtype=np.float64), np.asarray(got["summary"][k], dtype=np.float64), equal_nan=True), k
    names = [s_.name for s_ in scenes]
    assert _tree_digest(root, "plain", names) == _tree_digest(root, "rccl", names)


def test_bench_strong_scaling_two_ranks_on_one_gpu():
    """BASELINE configs[3] through bench.py itself: `--scenes-total 1201` shards ONE set of 1,201 scenes i mod W.  Two ranks (gloo rendezvous,
    both on cuda:0, small scenes) must report the same all-reduced pseudo-label mIoU over the same 1,201 scenes as one rank, with their
    own parity checks green -- the bench's rank logic on the hardware a test box has."""
    import json
    import subprocess
    common = ["--scenes-total", "1201", "--points", "3000", "--segments", "30", "--steps", "2", "--warmup", "1", "--repeats", "1",
              "--no-cpu-baseline", "--no-files", "--no-extras", "--batch", "64", "--parity-scenes", "8", "--backend", "gloo", "--groups", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1 and j2["scaling"] == "strong" and j1["scaling"] == "strong"
    assert j2["parity_check"]["ranks_equal"] and j1["parity_check"]["ranks_equal"]
    assert j2["pseudo_label_mIoU"]["scenes"] == j1["pseudo_label_mIoU"]["scenes"] == 1201 * 2               # the timed steps' scenes, all-reduced
    assert j2["pseudo_label_mIoU"]["semantic"] == j1["pseudo_label_mIoU"]["semantic"]
    assert j2["pseudo_label_mIoU"]["instance"] == j1["pseudo_label_mIoU"]["instance"]


def test_full_size_scenes_cross_the_rank_logic_and_the_synthetic_flag(tmp_path, weight_sets):
    """(a) `infer.py --synthetic N` writes the tree it then runs (SURVEY section 5's config row); (b) configs[3]'s sharding
    with FULL-SIZE scenes: four 150k-point / 1.5k-segment scenes written by that flag, then the same tree through two gloo ranks on cuda:0
    (two scenes each, `i mod 2`) -- every scene's 28 label files byte-identical to the one-process run, the all-reduced metrics equal."""
    import torch
    import torch.multiprocessing as mp
    from seggroup_amd import infer, weights
    root = str(tmp_path)
    common = ["--ins_infer", "--root", root, "--batch", "2", "--inflight", "4", "-j", "4"]
    ck = os.path.join(root, "checkpoints", "w1", "models")
    os.makedirs(ck)
    torch.save({"state_dict": weights.to_full_state_dict(weight_sets["ins_infer"])}, os.path.join(ck, "last.t7"))
    infer.main(["-n", "w1", "--world-size", "1", "--synthetic", "4"] + common)                 # writes dataset/ + runs W = 1
    names = [l.strip() for l in open(os.path.join(root, "dataset", "scannet", "scannetv2_train.txt"))]
    assert names == ["scene%04d_00" % i for i in range(4)]
    with pytest.raises(SystemExit):                                                         # a tree that exists is never overwritten
        infer.main(["-n", "w1", "--world-size", "1", "--synthetic", "4"] + common)
    log = open(os.path.join(root, "checkpoints", "w1", "run_infer.log")).read()
    assert "Wrote 4 synthetic scenes (150000 points / 1500 segments" in log and "==> Infer           Instance mIoU:" in log
    ck = os.path.join(root, "checkpoints", "w2", "models")
    os.makedirs(ck)
    torch.save({"state_dict": weights.to_full_state_dict(weight_sets["ins_infer"])}, os.path.join(ck, "last.t7"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker_full, args=(r, 2, root, port, "w2", q)) for r in range(2)]
    for p in procs:
        p.start()
    two = q.get(timeout=900)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    a, b = _tree_digest(root, "w1", names), _tree_digest(root, "w2", names)
    assert a == b, [n for n in names if a[n] != b[n]]
    assert two["n"] == 4
    V = np.load(os.path.join(root, "results", "w2", names[3], "ins_infer", "final.ins.npy")).shape[0]
    assert V >= 150000


def _rank_worker_full(rank, world, root, port, exp, q):
    sys.path.insert(0, ROOT)
    from seggroup_amd import infer
    args = infer.build_parser().parse_args(["-n", exp, "--ins_infer", "--root", root, "--backend", "gloo", "--port", str(port),
                                            "--batch", "2", "--inflight", "4", "-j", "4"])
    r = infer.run_worker(rank, world, args)
    if rank == 0:
        q.put({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in r.items()})


def test_eight_ranks_on_one_gpu_driver(tmp_path, golden_index, weight_sets):
    """nothing had run more than TWO ranks.  Eight processes (gloo rendezvous, all on cuda:0, each with its own engine,
    loader, writer pool and NUMA bind) over a 19-scene tree -- `i mod 8` leaves ranks with three and with two scenes -- must write byte-identical
    files to the one-process run and all-reduce to the same metric vector."""
    import torch
    import torch.multiprocessing as mp
    from seggroup_amd import infer, synthetic, weights
    root = str(tmp_path)
    n_scenes = 19
    scenes = []
    for i in range(n_scenes):
        if i == 0:
            e = golden_index["tiny_4k"]
            scenes.append(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]))
        else:
            scenes.append(synthetic.make_scene(3000 + 211 * i, 30 + 3 * i, 85000 + i, name=f"scene{i:04d}_00", **({"dup_frac": 0.05} if i % 4 == 0 else {})))
    synthetic.write_reference_tree(root, scenes)
    names = [s.name for s in scenes]
    for exp in ("w1", "w8"):
        ck = os.path.join(root, "checkpoints", exp, "models")
        os.makedirs(ck)
        torch.save({"state_dict": weights.to_full_state_dict(weight_sets["ins_infer"])}, os.path.join(ck, "last.t7"))
    one = infer.run_worker(0, 1, infer.build_parser().parse_args(
        ["-n", "w1", "--ins_infer", "--root", root, "--world-size", "1", "--batch", "5", "--inflight", "4", "-j", "2"]))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 8, root, port, "shard", "w8", q)) for r in range(8)]
    for p in procs:
        p.start()
    eight = q.get(timeout=900)
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    a, b = _tree_digest(root, "w1", names), _tree_digest(root, "w8", names)
    assert a == b, [n for n in names if a[n] != b[n]]
    g = load_golden("tiny_4k")
    from seggroup_amd import hip
    for nm in hip.LABEL_NAMES:
        assert np.array_equal(np.load(os.path.join(root, "results", "w8", names[0], "ins_infer", nm + ".npy")), g[f"ins.label.{nm}"]), nm
    assert eight["n"] == one["n"] == n_scenes
    for k in one:
        if k not in ("elapsed_s", "startup_s", "first_batch"):
            assert np.array_equal(np.asarray(one[k]), np.asarray(eight[k]), equal_nan=True), k


def test_bench_eight_ranks_on_one_gpu_and_the_comm_field():
    """`bench.py --gpus 8` as the round-end driver launches it (torch.distributed.run, eight ranks) on the one GPU a test box has (gloo; small
    scenes): the line's `comm` object must show what the collective layer saw -- eight ranks in the communicator, an all-reduce of ones = 8 --
    and the strong-scaling set must reduce to the same pseudo-label mIoU as at one rank.  At N = 1 the field is there too, over a one-rank RCCL
    communicator made for the probe."""
    import json
    import subprocess
    common = ["--scenes-total", "97", "--points", "3000", "--segments", "30", "--steps", "2", "--warmup", "1", "--repeats", "1",
              "--no-cpu-baseline", "--no-files", "--no-extras", "--batch", "8", "--parity-scenes", "4", "--groups", "2", "--per-group", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    eight = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo"] + common,
                           capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert eight.returncode == 0, eight.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl"] + common, capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    j8 = json.loads([l for l in eight.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j8["n_gpus"] == 8 and j8["scaling"] == "strong" and j8["parity_check"]["ranks_equal"] and j1["parity_check"]["ranks_equal"]
    assert j8["comm"]["backend"] == "gloo" and j8["comm"]["world_size"] == 8 and j8["comm"]["allreduce_of_ones"] == 8.0, j8["comm"]
    assert j1["comm"]["backend"] == "nccl" and j1["comm"]["world_size"] == 1 and j1["comm"]["allreduce_of_ones"] == 1.0 and j1["comm"]["librccl_mapped"], j1["comm"]
    assert j8["pseudo_label_mIoU"]["scenes"] == j1["pseudo_label_mIoU"]["scenes"] == 97 * 2
    assert j8["pseudo_label_mIoU"]["semantic"] == j1["pseudo_label_mIoU"]["semantic"]
    assert j8["pseudo_label_mIoU"]["instance"] == j1["pseudo_label_mIoU"]["instance"]


@pytest.mark.parametrize("mode", ["ins_infer", "sem_infer"])
def test_run_infer_log_equals_the_reference_transcript(tmp_path, golden_index, weight_sets, mode):
    """`tests/golden/transcript_<mode>.log` is what the reference's own `infer()` logged (tools/capture_transcript.py: the
    unmodified infer.py:127-190 over an eight-scene tree, world size 1, its DistributedSampler's order).  The same tree through `seggroup_amd.infer`
    on the GPU (`--sampler reference`, the packed fast path): `run_infer.log` from the first `Infer(` line on is the transcript BYTE FOR BYTE --
    every running mIoU / accuracy, the `==> Infer` line, both per-class tables with their `nan%` rows -- and the header carries the reference's
    parameter count."""
    import json
    import torch
    from seggroup_amd import infer, synthetic, weights
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "transcript.json")))
    root = str(tmp_path)
    scenes = [synthetic.make_scene(s["n"], s["s"], s["seed"], name=s["name"], **s["kw"]) for s in meta["scenes"]]
    synthetic.write_reference_tree(root, scenes)
    ck = os.path.join(root, "checkpoints", "exp", "models")
    os.makedirs(ck)
    torch.save({"state_dict": weights.to_full_state_dict(weight_sets[mode])}, os.path.join(ck, "last.t7"))
    args = infer.build_parser().parse_args(["-n", "exp", f"--{mode}", "--root", root, "--world-size", "1", "--sampler", "reference", "--batch", "3",
                                            "--inflight", "4", "-j", "2"])
    infer.run_worker(0, 1, args)
    text = open(os.path.join(root, "checkpoints", "exp", "run_infer.log")).read()
    at = text.index("Infer(0001/")
    want = open(os.path.join(ROOT, "tests", "golden", f"transcript_{mode}.log")).read()
    assert text[at:] == want
    head = text[:at].splitlines()
    assert "Network parameters: %d" % meta[mode]["network_parameters"] in head                 # infer.py:92
    assert any(l.startswith("Load model from ") and l.endswith("checkpoints/exp/models/last.t7") for l in head)      # infer.py:119


def test_configs3_full_size_1201_scenes_over_eight_ranks(tmp_path):
    """BASELINE.json configs[3] AT FULL SIZE on the hardware a test box has: the 1,201-scene set (150k points / 1.5k segments each, seeds 40000 + i) sharded `i mod 8`
    over EIGHT ranks of `bench.py` as the round-end driver launches it (torch.distributed.run; gloo rendezvous, all ranks on cuda:0, each with its own engine), against the
    same set on one rank: the all-reduced pseudo-label mIoU over the 1,201 scenes is the same number, every rank's parity check (8 scenes of its last batch against the
    single pipeline) is green, the communicator saw eight ranks.  The scenes are generated once into a cache both runs read."""
    import json
    import subprocess
    cache = str(tmp_path / "scenes")
    common = ["--scenes-total", "1201", "--steps", "1", "--warmup", "0", "--repeats", "1", "--no-cpu-baseline", "--no-files", "--no-extras", "--batch", "64",
              "--parity-scenes", "8", "--no-oos", "--scene-cache", cache]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "gloo", "--gen-workers", "48"] + common, capture_output=True, text=True, timeout=1500,
                         env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    assert len(os.listdir(cache)) == 1201
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    eight = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--groups", "4", "--gen-workers", "6"] + common,
                           capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert eight.returncode == 0, eight.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j8 = json.loads([l for l in eight.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["scaling"] == j8["scaling"] == "strong" and j8["n_gpus"] == 8 and "1201 distinct synthetic scenes" in j8["config"]["workload"]
    assert j1["parity_check"]["ranks_equal"] and j8["parity_check"]["ranks_equal"]
    assert j8["comm"]["world_size"] == 8 and j8["comm"]["allreduce_of_ones"] == 8.0
    assert j1["pseudo_label_mIoU"]["scenes"] == j8["pseudo_label_mIoU"]["scenes"] == 1201
    assert j1["pseudo_label_mIoU"]["semantic"] == j8["pseudo_label_mIoU"]["semantic"] and j1["pseudo_label_mIoU"]["instance"] == j8["pseudo_label_mIoU"]["instance"]
