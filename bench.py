#!/usr/bin/env python3
"""Headline benchmark: pseudo-label scenes/sec on synthetic ScanNet-shaped scenes (150k points /
1.5k segments), SegModel.forward in ins_infer mode through the C ABI.

    python bench.py [--gpus N --steps K --warmup W]

N > 1 without a launcher: this process starts `torch.distributed.run` with N ranks itself (before it
touches the GPU) and relays rank 0's JSON line; under torchrun (RANK / WORLD_SIZE set) it is one rank.

A "step" = one batch of `--batch` (64: BASELINE.json configs[2]) distinct scenes per GPU (weak scaling:
per-GPU work is fixed as N grows; scenes are independent, no collective in the data path -- one RCCL
all-reduce of the metric accumulators at the end).  `--scenes-total T` instead shards ONE
set of T scenes over the ranks (scene i -> rank i mod W; configs[3] is T = 1201) and a step is one pass
over the rank's shard: strong scaling.  Scenes are staged in HBM before the timed region; the timed
region covers everything SegModel.forward does for a scene (all kernels, the serial host grouping, D2H
of the 14 label vectors and metrics) except writing the label files (reported separately as
`with_label_files_scenes_per_s`, through the asynchronous native writer pool).

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job scenes/s, plus
  roofline     - the dominant MFMA kernel, k_edgeconv<S2X>: the 16-bit MFMA work the launch EXECUTES
                  / the duration of the batched launch with ONE engine group on the
                 GPU ("solo batched", HIP events on the group's stream, measured in this process after the
                 timed region; `profiles/rNN_solo_batched_kernel_stats.csv` is the rocprofv3 view of the same
                 configuration) / the 2.5 PF dense 16-bit MFMA peak; the algorithmic fp32 contraction and
                 the whole-GPU figure as secondary keys;
  cpu_baseline - the NumPy oracle ("port", faithful per-edge loops) timed on this box's host cores on a
                 bounded sample (one scene of the same workload, repeated), rank 0 at N=1 only;
  parity_check - after the timed region, label digests of EVERY scene of the last batch against the same
                 scenes through a single default-stream pipeline;
  extra        - strong_1201 (BASELINE configs[3] at W = 1), latency (configs[1] / configs[4]: one scene alone),
                 the ScanNet-shaped profile, the training step.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight stream (HIP's default of 4 makes streams share queues);
# must be set before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak (64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz)
MFMA_BF16_PEAK_TF = 2500.0     # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md; 2495 TF measured)
# what the EdgeConv launches EXECUTE per edge row (one of the 20 neighbour slots of a point): fp32 operands split into 16-bit
# pieces that meet on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation.  An MFMA of 32x32x16 is
# 2 * 32 * 32 * 16 flop for 32 rows; CONV1_MFMAS = MFMAs per neighbour slot and 32-row tile for the 9-deep conv1 (two 32-channel
# output tiles; the three products of the fp16 x 2 split packed along K into two MFMAs per tile: round 3 -- 12 with bf16 x 3 in round 2),
# conv2 = three fp16 products of 64x64 per row (hi*hi + hi*lo + lo*hi).
CONV1_MFMAS_PER_SLOT = 4
CONV1_EXEC_FLOP_PER_ROW = CONV1_MFMAS_PER_SLOT * 2 * 32 * 32 * 16 // 32
S1X_EXECUTED_FLOP_PER_ROW = CONV1_EXEC_FLOP_PER_ROW
S2X_EXECUTED_FLOP_PER_ROW = 3 * 2 * 64 * 64 + CONV1_EXEC_FLOP_PER_ROW
# VALU issue roof, MEASURED (tools/micro/issue_rates.hip, 8 waves per SIMD, every SIMD busy): one wave64 instruction costs a SIMD 1.96-2.07 ns
# for v_max_f32 / v_max_f64 / v_cvt_pk_f16_f32 / v_pk_fma_f32 alike (1.37 ns for v_fma_f32) -- about four cycles at the clock the chip
# sustains, not the two of the SIMD-32 data path (round 2 priced the kNN against 1,229 G instructions/s).  1024 SIMDs / 2.0 ns:
VALU_ISSUE_NS = 2.0
VALU_PEAK_GINST = 1024 / VALU_ISSUE_NS
PROFILE_TAG = "r06"
DTYPE = "f32 (fp32 accumulate; operands split into 2 x fp16 pieces on v_mfma_f32_32x32x16_f16; kNN / FPS scores in exact fp32 order)"


def kernel_model(n_points: int, k: int = 20):
    """Hot kernels: the engine stages that time them (HIP events), launches per scene, and per SCENE-launch the ALGORITHMIC work
    (fp32 contraction flops / bytes) and the EXECUTED 16-bit MFMA flops."""
    n = float(n_points)
    c1, c12 = 2.0 * k * n * (18 * 64), 2.0 * k * n * (18 * 64 + 64 * 64)
    return {
        # name: (stages, launches per scene, bound, algorithmic units, unit, executed MFMA flop)
        "k_edgeconv<S2X>": (["kernel.l3.edgeconv"], 1, "mfma", c12, "TFLOP/s", S2X_EXECUTED_FLOP_PER_ROW * k * n),
        "k_edgeconv<S1X>": (["kernel.l2.edgeconv"], 1, "mfma", c1, "TFLOP/s", S1X_EXECUTED_FLOP_PER_ROW * k * n),
        # kNN: reads [N,4] f32, writes [N,20] i32; VALU-bound (selection): priced against the VALU issue roof from the PMC pass
        "k_cluster_knn_sorted": (["l2.knn", "l3.knn"], 2, "valu", 96.0 * n, "GB/s", 0.0),
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80, help="timed steps (80 x 64 scenes: a timed region of > 2 s)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each: `value` is the FIRST (the contract's K steps), "
                                                            "the others are reported as repeat_values (run-to-run spread)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps in front of the timed region (clocks and caches settle over the first ~0.2 s)")
    ap.add_argument("--batch", type=int, default=64, help="distinct scenes per GPU per step (BASELINE.json configs[2]: 64)")
    ap.add_argument("--scenes-total", type=int, default=0,
                    help="strong scaling: ONE set of this many scenes sharded i mod W over the ranks (configs[3]: 1201); "
                         "a step = one pass over the rank's shard in batches of --batch")
    ap.add_argument("--groups", type=int, default=14, help="engine groups per GPU (host thread + HIP stream each).  10 through round 5 and most of round 6; with the label vectors off the CUs "
                                                              "(csrc/sdma.cpp) more groups pay again: 10 / 14 / 16 / 18 / 20 x 8 = 3,420-3,443 / 3,444-3,497 / 3,375-3,504 / 3,295-3,414 / 3,117-3,224 scenes/s "
                                                              "(beyond 16 the streams share hardware queues: GPU_MAX_HW_QUEUES=16)")
    ap.add_argument("--per-group", type=int, default=8, help="scenes a group advances in lock-step through batched launches")
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--segments", type=int, default=1500)
    ap.add_argument("--seg-profile", default="voronoi", help="synthetic segment-size profile: voronoi (SURVEY 8d recipe) | scannet (heavy-tailed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline leg")
    ap.add_argument("--cpu-curve-full", action="store_true", help="CPU baseline leg: also time the port on 64 threads and on ALL host cores (minutes on a 256-core host)")
    ap.add_argument("--no-files", action="store_true", help="skip the separate with-files measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra leg (scenes/s on the ScanNet-shaped segment profile)")
    ap.add_argument("--extra-train", type=int, default=8, help="training steps timed in the extra leg (rank 0, N = 1; 0 = skip)")
    ap.add_argument("--extra-scannet", type=int, default=64, help="scenes of the ScanNet-shaped profile in the extra leg (rank 0, N = 1)")
    ap.add_argument("--writer-threads", type=int, default=12, help="native writer threads for the with-files leg (tmpfs, txt + npy: 8 threads 1,690, 12 2,020, 16 1,830, 32 1,250-1,390 scenes/s once the text comes from a table: more threads only contend for memory bandwidth)")
    ap.add_argument("--gen-workers", type=int, default=0, help="scene generator processes (0 = min(16, cores))")
    ap.add_argument("--numa", default="auto", choices=["auto", "off"], help="auto = bind every rank's process (engine groups, writer pool) to the CPUs of its GPU's NUMA node")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for CPU rehearsals of the reduction)")
    ap.add_argument("--parity-scenes", type=int, default=64, help="scenes of the last batch re-run on a single pipeline and compared (default: all 64); 0 = no parity legs at all "
                                                                  "(profiled runs: the out-of-step leg launches the batched kernels with ONE scene each, which dilutes per-launch profile averages)")
    ap.add_argument("--no-oos", action="store_true", help="skip the out-of-step parity leg (16 groups x 1 scene) only")
    ap.add_argument("--label-transfer", default="full", choices=["full", "tables"], help="full = the 14 label vectors of every scene cross PCIe inside the timed region (what SegModel.forward "
                                                                                         "returns: the metric's definition); tables = only the [14,S] tables do (an EXPERIMENT: what the label copies cost the engine)")
    ap.add_argument("--engine-timing", type=int, default=1, choices=[0, 1], help="1 = the engine records its stage events (~24 per group super-step) INSIDE the timed region too "
                                                                                   "(stage_ms_in_timed_region); 0 = none there, and the stage times under load come from an extra untimed pass")
    ap.add_argument("--profile", action="store_true", help="roctx ranges around the bench's legs and, inside the engine's group threads, around every phase and stage "
                                                           "(rocprofv3 --kernel-trace --marker-trace --stats -- python3 bench.py --profile ...; tools/prof_ranges.sh)")
    ap.add_argument("--extra-strong", type=int, default=1201, help="extra leg (rank 0, N = 1): ONE pass over this many distinct scenes = BASELINE configs[3] "
                                                                    "at W = 1 (0 = skip)")
    ap.add_argument("--extra-stress", type=int, default=500000, help="extra leg: single-scene latency of a scene with this many points / 100 (0 = skip)")
    ap.add_argument("--generate-only", action="store_true", help="fill --scene-cache (worker pool, no GPU call) and exit")
    ap.add_argument("--scene-cache", default="", help="directory of generated scenes (.npz per scene): read when present, written otherwise; "
                                                      "profiled runs use it with --gen-workers 1 so that the profiled process spawns nothing")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no torchrun around it
# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv) -> int:
    """Start N ranks as CHILD processes (torch.distributed.run) before this process has made any HIP call, relay
    their output, return the children's exit code.  Never re-execs: a parent that has initialised the GPU must not."""
    import torch  # importing torch and counting devices does not initialise the GPU on this image
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested but only {n_dev} HIP device(s) are visible; refusing to run a "
              f"{n_dev}-GPU measurement under an n_gpus={args.gpus} label", file=sys.stderr)
        return 2
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SG_BENCH_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        return 3
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# scene generation in worker processes (NumPy / SciPy only: nothing there touches the GPU)
# ------------------------------------------------------------------------------------------------------------------
def _gen_job(job):
    points, segments, seed, profile, cache = job
    from seggroup_amd import synthetic
    path = os.path.join(cache, f"scene_{profile}_{points}_{segments}_{seed}.npz") if cache else None
    if path and os.path.exists(path):
        z = np.load(path)
        return synthetic.Scene(name=str(z["name"]), **{k: z[k] for k in ("data", "weak_label", "seg", "adj", "unmap", "gt")})
    kw = {} if profile == "voronoi" else {"seg_profile": profile}
    sc = synthetic.make_scene(points, segments, seed, **kw)
    if path:
        os.makedirs(cache, exist_ok=True)
        tmp = path + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, name=sc.name, data=sc.data, weak_label=sc.weak_label, seg=sc.seg, adj=sc.adj, unmap=sc.unmap, gt=sc.gt)
        os.replace(tmp, path)
    return sc


def generate_scenes(jobs, workers):
    """Ordered iterator over host scenes; the pool's processes are started (spawn context) by the submits below,
    i.e. before the caller initialises the GPU.  workers <= 1 (profiled runs: nothing may be spawned under rocprofv3)
    generates -- or reads from --scene-cache -- inline."""
    if workers <= 1 or len(jobs) < 4:
        return (_gen_job(j) for j in jobs), None
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    pool = ProcessPoolExecutor(max_workers=min(workers, len(jobs)), mp_context=mp.get_context("spawn"))
    futs = [pool.submit(_gen_job, j) for j in jobs]
    return (f.result() for f in futs), pool


def label_digest(res) -> str:
    h = hashlib.sha256()
    for i in range(res.n_vectors):
        h.update(np.ascontiguousarray(res.labels[i]).tobytes())
    h.update(np.ascontiguousarray(res.iou_sem).tobytes())
    h.update(np.ascontiguousarray(res.iou_ins).tobytes())
    h.update(np.nan_to_num(np.ascontiguousarray(res.acc), nan=-1.0).tobytes())
    h.update(np.asarray(res.trace, dtype=np.int32).tobytes())
    return h.hexdigest()


def reduce_accumulators(vec: np.ndarray, world: int, backend: str, dev=None) -> np.ndarray:
    """The path's only collective: one all-reduce of the float64 metric accumulators
    [I_sem 40 | U_sem 40 | I_ins 40 | U_ins 40 | acc 4 | scenes 1]."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return vec
    t = torch.from_numpy(vec.copy())
    if backend == "nccl":
        t = t.to(dev)
    dist.all_reduce(t)
    return t.cpu().numpy()


def comm_probe(world: int, backend: str, dev) -> dict:
    """What the collective layer saw: the backend, the communicator's own world size, the result of an all-reduce of
    ones over it (= the number of ranks that took part) and whether librccl is mapped into this process.  Called by EVERY rank (the all-reduce is
    a collective); at N = 1 without a launcher a one-rank communicator of the same backend is made for the probe and destroyed again."""
    import socket

    import torch
    import torch.distributed as dist
    info = {"backend": backend, "world_size": None, "allreduce_of_ones": None, "librccl_mapped": False}
    own = False
    try:
        if not dist.is_initialized():
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            kw = {"device_id": dev} if backend == "nccl" else {}
            dist.init_process_group(backend=backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
            own = True
        t = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        info["world_size"] = dist.get_world_size()
        info["allreduce_of_ones"] = float(t.item())
        info["expected"] = world
    except Exception as e:                                  # the probe must not take the bench line with it
        info["error"] = repr(e)[:200]
    finally:
        if own:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    try:
        with open("/proc/self/maps") as f:
            info["librccl_mapped"] = any("librccl" in ln for ln in f)
    except OSError:
        pass
    return info


def digest_of_digests(results) -> str:
    h = hashlib.sha256()
    for r in results:
        h.update(label_digest(r).encode())
    return h.hexdigest()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args, argv))

    # stdout carries ONE line, the result: everything else a rank's process writes to fd 1 -- Python prints, and C stdio of the libraries (RCCL
    # prints its version banner there, buffered until exit, i.e. BEHIND the JSON line) -- goes to stderr from here on
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with matching values, or let bench.py start the ranks itself)")

    # ---- synthetic input, generated in worker processes BEFORE this process initialises the GPU ----
    t_gen = time.time()
    cores = os.cpu_count() or 1
    extras_on = not (args.no_extras or world > 1 or args.seg_profile != "voronoi" or args.scenes_total > 0)
    n_strong = max(0, args.extra_strong) if extras_on else 0
    workers = args.gen_workers or max(1, min(64 if n_strong else 16, cores // (2 * max(world, 1)) or 1))
    cache = args.scene_cache
    if args.scenes_total > 0:
        mine = list(range(rank, args.scenes_total, world))              # scene i -> rank i mod W
        jobs = [(args.points, args.segments, 40000 + i, args.seg_profile, cache) for i in mine]
    else:
        jobs = [(args.points, args.segments, 30000 + 1000 * rank + i, args.seg_profile, cache) for i in range(args.batch)]
    n_main = len(jobs)
    n_scannet = max(0, args.extra_scannet) if extras_on else 0
    jobs += [(args.points, args.segments, 70100 + i, "scannet", cache) for i in range(n_scannet)]
    jobs += [(args.points, args.segments, 40000 + i, args.seg_profile, cache) for i in range(n_strong)]     # the set `--scenes-total` shards
    n_stress = 1 if (extras_on and args.extra_stress > 0) else 0
    jobs += [(args.extra_stress, args.extra_stress // 100, 50005, "voronoi", cache)] * n_stress
    scene_iter, pool = generate_scenes(jobs, workers)
    if args.generate_only:
        n_gen = sum(1 for _ in scene_iter)
        if pool is not None:
            pool.shutdown()
        print(f"bench.py: {n_gen} scenes in {args.scene_cache or '(no cache directory given)'} ({time.time() - t_gen:.1f} s)", file=sys.stderr)
        return

    if args.profile:
        os.environ["SG_ROCTX"] = "1"                            # read when the library loads (engine group threads push their own ranges)
    import torch
    import torch.distributed as dist

    from seggroup_amd import hip, weights
    from seggroup_amd.scene import DeviceScene

    hip.require_device()
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} HIP device(s): one rank per GPU")
    local = local % max(n_dev, 1)                       # several ranks on one GPU only in gloo rehearsals
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from seggroup_amd.numa import bind_to_gpu_node
    numa_info = bind_to_gpu_node(local, args.numa)          # before the engine's group threads and the writer pool exist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
    host_scene0 = None
    scenes = []
    for i, s in enumerate(scene_iter):                  # upload as they arrive; host copies are dropped
        if i == 0:
            host_scene0 = s
        scenes.append(DeviceScene.from_synthetic(s, device=dev))
    if pool is not None:
        pool.shutdown()
    gen_s = time.time() - t_gen
    stress_scenes = scenes[len(scenes) - n_stress:] if n_stress else []
    scenes = scenes[:len(scenes) - n_stress] if n_stress else scenes
    strong_scenes, scenes = scenes[n_main + n_scannet:], scenes[:n_main + n_scannet]
    extra_scenes, scenes = scenes[n_main:], scenes[:n_main]

    from seggroup_amd.model import Engine, Pipeline
    from seggroup_amd.hip import roctx_range as rng            # no-ops unless --profile (SG_ROCTX) is on
    every = scenes + extra_scenes + strong_scenes
    caps = (max(s.N for s in every), max(s.S for s in every), max(s.E0 for s in every), max(s.V for s in every))
    # stage timing: a handful of HIP events per batched launch sequence (per group of scenes, not per scene)
    runner = Engine(W, caps, groups=args.groups, per_group=args.per_group, device=dev, timing=args.engine_timing, label_transfer=args.label_transfer)
    acc = {"iou_sem": np.zeros(80), "iou_ins": np.zeros(80), "acc": np.zeros(4), "n": 0}
    batches = [scenes[k:k + args.batch] for k in range(0, len(scenes), args.batch)]

    def run_batches(bl, k, sink=None):
        """k passes over the batches `bl`, queued two ahead (submit k+2, then wait for k) as a driver with a stream of scenes does:
        the engine's groups never drain between two batches.  Every batch is waited for and its results handed to `sink` (which
        must consume them before the third submit after theirs: the label buffers are a ring) inside the call."""
        last, pending = None, []
        for _ in range(k):
            for b in bl:
                pending.append(runner.submit(b, hip.MODE_INS_INFER))
                if len(pending) > 2:
                    last = runner.wait(pending.pop(0))
                    if sink:
                        sink(last)
        while pending:
            last = runner.wait(pending.pop(0))
            if sink:
                sink(last)
        return last

    def record(res):
        for r in res:
            acc["iou_sem"] += r.iou_sem.reshape(-1)
            acc["iou_ins"] += r.iou_ins.reshape(-1)
            acc["acc"] += np.nan_to_num(r.acc)
            acc["n"] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, sink):
        barrier()
        t0_ = time.perf_counter()
        last_ = run_batches(batches, k, sink)
        barrier()
        el = time.perf_counter() - t0_
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, last_

    runner.profile(enable=True)
    with rng("warmup"):
        run_batches(batches, args.warmup)
    runner.reset_stage_stats()
    with rng("timed_region"):
        elapsed, last_results = timed(args.steps, record)       # THE timed region: exactly --steps steps
    engine_profile = {k_: round(v, 3) for k_, v in runner.profile().items()}
    mean_ms = runner.mean_stage_ms()
    scenes_per_step = args.scenes_total if args.scenes_total > 0 else world * args.batch
    value = scenes_per_step * args.steps / elapsed
    # labels of the last timed batch, digested before anything else is submitted (their buffers are a ring)
    last_batch = batches[-1]
    n_par = min(args.parity_scenes, len(last_batch))
    batch_digests = [label_digest(last_results[i]) for i in range(n_par)]
    batch_trace0 = list(last_results[0].trace)
    repeat_values = [round(value, 3)]
    for _ in range(max(0, args.repeats - 1)):                   # run-to-run spread; `value` stays the first region
        el, _ = timed(args.steps, None)
        repeat_values.append(round(scenes_per_step * args.steps / el, 3))

    if not args.engine_timing:
        # stage times under the bench's load from an extra, untimed pass with the events on (the timed regions above ran without them)
        runner.set_timing(1)
        runner.reset_stage_stats()
        run_batches(batches, max(2, args.steps // 8))
        mean_ms = runner.mean_stage_ms()
        runner.set_timing(0)
    vec = reduce_accumulators(np.concatenate([acc["iou_sem"], acc["iou_ins"], acc["acc"], [acc["n"]]]), world, args.backend, dev)
    # What the collective layer saw.  With ranks (a process group exists since start-up) every rank takes part here.  At N = 1 the probe has to
    # BUILD (and remove) a communicator; it runs LAST there, behind every measured leg, so that no leg's number can depend on it (a precaution:
    # an A / B of the timed region with a communicator built up front showed no difference).
    comm = comm_probe(world, args.backend, dev) if world > 1 else None

    # ---- parity of the concurrent path: EVERY scene of the LAST timed batch vs a single default-stream pipeline ----
    solo = Pipeline(W, *caps, stream=None, device=dev)
    solo.set_timing(0)
    with rng("parity.single_pipeline"):
        solo_digests = [label_digest(solo.forward(last_batch[i], hip.MODE_INS_INFER)) for i in range(n_par)]
    parity_ok = batch_digests == solo_digests
    # ---- the same scenes once more with the groups OUT OF STEP: sixteen groups of one scene each, so that waves of different kernels share the
    # SIMDs (the tail of every driver run; round 5 found results that depended on the run only there: DESIGN.md 5e).  Untimed. ----
    oos_runs, oos_wrong = (0 if (n_par == 0 or args.no_oos) else 3), 0
    if oos_runs:
        with rng("parity.out_of_step"):
            oos = Engine(W, caps, groups=16, per_group=1, device=dev, timing=0)
            for _ in range(oos_runs):
                got = [label_digest(r_) for r_ in oos.run(list(last_batch[:n_par]), hip.MODE_INS_INFER)]
                oos_wrong += sum(1 for a_, b_ in zip(got, solo_digests) if a_ != b_)
            oos.close()
    parity_ok = parity_ok and oos_wrong == 0
    ok = torch.tensor([1.0 if parity_ok else 0.0], dtype=torch.float64)
    if world > 1:
        if args.backend == "nccl":
            ok = ok.to(dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    parity_all = bool(ok.item() == 1.0)

    rc = 0
    if rank == 0:
        model = kernel_model(args.points)

        def per_kernel(ms):
            return {kn: sum(ms.get(st, 0.0) for st in m[0]) / m[1] for kn, m in model.items()}        # ms per scene-launch

        in_region = per_kernel(mean_ms)
        # ---- "solo batched": ONE group of --per-group scenes alone on the GPU, the same batched launches, HIP events on the group's
        # stream.  Under the timed region's load the groups' launches overlap and a launch's duration includes the time it shares the
        # device, so the roofline is taken from this configuration (rocprofv3 view: profiles/rNN_solo_batched_kernel_stats.csv).
        sb = Engine(W, caps, groups=1, per_group=args.per_group, device=dev, timing=1)
        sb.run(scenes[:args.per_group], hip.MODE_INS_INFER)
        sb.reset_stage_stats()
        with rng("solo_batched"):
            for k0 in range(0, len(scenes) - args.per_group + 1, args.per_group):       # full launches only
                sb.run(scenes[k0:k0 + args.per_group], hip.MODE_INS_INFER)
        sb_ms = sb.mean_stage_ms()
        sb.close()
        solo_b = per_kernel(sb_ms)
        sb_sum = sum(v for k_, v in sb_ms.items() if k_.count(".") <= 1)          # names with two dots are sub-passes of a stage

        pmc = {}
        ppath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_kernels.json")
        if os.path.exists(ppath) and args.points == 150000:
            pmc = json.load(open(ppath))
        dom = "k_edgeconv<S2X>"                                  # the kernel with the most MFMA work
        ex_all = sum(m[5] for m in model.values())               # executed 16-bit MFMA flop per scene (S1X + S2X)
        alg_all = sum(m[3] for m in model.values() if m[2] == "mfma")
        t_dom = solo_b.get(dom, 0.0) * 1e-3
        m_dom = model[dom]
        ach = m_dom[3] / t_dom / 1e12 if t_dom > 0 else 0.0       # ALGORITHMIC: 2 x 20 x N x (18 x 64 + 64 x 64) flop per scene-launch
        ach_exec = m_dom[5] / t_dom / 1e12 if t_dom > 0 else 0.0
        kernels = {}
        for kn, m in model.items():
            t = solo_b.get(kn, 0.0) * 1e-3
            if t <= 0:
                continue
            e = {"ms_per_scene_launch_solo_batched": round(t * 1e3, 4), "ms_per_scene_launch_in_timed_region": round(in_region.get(kn, 0.0), 4),
                 "launches_per_scene": m[1]}
            if m[2] == "mfma":
                e.update({"algorithmic_tflops": round(m[3] / t / 1e12, 1), "frac_of_16bit_mfma_peak": round(m[3] / t / 1e12 / MFMA_BF16_PEAK_TF, 4),
                          "executed_tflops": round(m[5] / t / 1e12, 1), "executed_frac_of_16bit_mfma_peak": round(m[5] / t / 1e12 / MFMA_BF16_PEAK_TF, 4)})
            kernels[kn] = e
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TF, 4),
                    "traffic": (pmc.get("hbm_bytes_per_scene_launch", {}) or {}).get(dom), "traffic_source": pmc.get("configuration"),
                    "basis": "ALGORITHMIC flop of one scene's share of the batched launch -- 2 x 20 x N x (18 x 64 + 64 x 64), the fp32 contraction of "
                             "MLP3 (SURVEY.md 8d) -- / the launch's duration with one engine group alone on the GPU (HIP events on the group's stream, after "
                             "the timed region; profiles/%s_solo_batched_kernel_stats.csv is rocprofv3's view of the same configuration) / the dense "
                             "16-bit MFMA peak.  The kernel EXECUTES 2.73x that (fp32 operands cut into two fp16 pieces, three products, fp32 accumulate: "
                             "fp32-emulated, DESIGN.md section 5): `executed` below" % PROFILE_TAG,
                    "executed": {"tflops": round(ach_exec, 1), "frac_of_16bit_mfma_peak": round(ach_exec / MFMA_BF16_PEAK_TF, 4),
                                 "flop_per_scene_launch": m_dom[5], "over_algorithmic": round(m_dom[5] / m_dom[3], 3)},
                    "executed_flop_per_scene_launch": m_dom[5], "algorithmic_flop_per_scene_launch": m_dom[3],
                    "ms_per_scene_launch": round(t_dom * 1e3, 4), "launches_per_scene": 1,
                    "algorithmic_vs_fp32_mfma_peak_157tf": round(m_dom[3] / t_dom / 1e12 / MFMA_F32_PEAK_TF, 3) if t_dom > 0 else None,
                    "whole_gpu": {"algorithmic_tflops": round(alg_all * value / world / 1e12, 1),
                                  "frac_of_16bit_mfma_peak": round(alg_all * value / world / 1e12 / MFMA_BF16_PEAK_TF, 4),
                                  "executed_mfma_tflops": round(ex_all * value / world / 1e12, 1),
                                  "executed_flop_per_scene": ex_all, "algorithmic_flop_per_scene": alg_all},
                    "kernels": kernels,
                    # consistency: the kernels' own time for a step's scenes must fit inside the step (overlapped in-region durations did not)
                    "check_kernel_time_fits_step": {"edgeconv_and_knn_ms_per_scene_x_scenes_per_step":
                                                    round(sum(solo_b.get(kn, 0.0) * m[1] for kn, m in model.items()) * scenes_per_step / world, 3),
                                                    "ms_per_step": round(elapsed / args.steps * 1e3, 3)},
                    "solo_batched_stage_sum_ms_per_scene": round(sb_sum, 4),        # stage intervals of one group alone: kernels + its host gaps
                    "stage_ms_solo_batched": {k_: round(v, 4) for k_, v in sb_ms.items() if v > 0},
                    "stage_ms_in_timed_region": {k_: round(v, 4) for k_, v in mean_ms.items() if v > 0}}
        # VALU roof of the in-cluster kNN: SQ_INSTS_VALU per scene-launch (PMC pass, committed) / (1024 SIMDs x 2.4 GHz / 2 cycles)
        vmap = pmc.get("valu_insts_per_scene_launch", {}) or {}
        vi = sum(vmap.get(k_, 0) for k_ in ("k_cluster_knn_sorted<unseeded>", "k_cluster_knn_sorted<seeded>"))
        if vi and solo_b.get("k_cluster_knn_sorted", 0) > 0:
            ms2 = 2.0 * solo_b["k_cluster_knn_sorted"]                                             # both launches of a scene
            hb = pmc.get("hbm_bytes_per_scene_launch", {}) or {}
            roofline["knn_valu"] = {"valu_insts_per_scene": vi, "peak_ginst_per_s": VALU_PEAK_GINST,
                                    "peak_basis": "measured issue cost of a wave64 VALU instruction per SIMD with 8 waves per SIMD: ~2.0 ns (tools/micro/issue_rates.hip)",
                                    "ms_per_scene_both_launches_solo_batched": round(ms2, 4),
                                    "valu_frac": round(vi / (ms2 * 1e-3) / 1e9 / VALU_PEAK_GINST, 4),
                                    "hbm_bytes_per_scene_launch": {k_: hb.get(k_) for k_ in ("k_cluster_knn_sorted<unseeded>", "k_cluster_knn_sorted<seeded>")},
                                    # per point: the sorted operand (16 B) + its member position (4) + its point id (4) read, the 20-entry table written (80); the layer-2
                                    # launch writes the table a second time in point ids for the next layer's seeds (80), the seeded launch reads that row (80) and the
                                    # seeds' records (16 B per point of the array, gathered out of L2).  (96 B through round 5: the seed tables were not counted.)
                                    "algorithmic_bytes_per_scene_launch": {"k_cluster_knn_sorted<unseeded>": 184.0 * args.points, "k_cluster_knn_sorted<seeded>": 200.0 * args.points},
                                    "source": f"profiles/{PROFILE_TAG}_pmc_kernels.json (SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE of the two kNN launches, separate PMC passes)"}
        # the WHOLE job against the same issue roof: every batched kernel's SQ_INSTS_VALU per scene (launches per scene from the PMC pass's own launch
        # counts, `k_mlp1_apply_b` runs once per scene) x the scenes/s of the timed region / the issue rate of 1,024 SIMDs
        rawk = pmc.get("per_kernel_raw", {}) or {}
        base_l = (rawk.get("k_mlp1_apply_b", {}) or {}).get("launches", 0)
        if vmap and base_l:
            per_scene = sum(v_ * rawk.get(k_, {}).get("launches", base_l) / base_l for k_, v_ in vmap.items())
            roofline["whole_job_valu"] = {"valu_insts_per_scene": int(per_scene), "scenes_per_s_per_gpu": round(value / world, 1),
                                          "achieved_ginst_per_s": round(per_scene * value / world / 1e9, 1), "peak_ginst_per_s": VALU_PEAK_GINST,
                                          "frac": round(per_scene * value / world / 1e9 / VALU_PEAK_GINST, 4),
                                          "scenes_per_s_at_the_roof": round(VALU_PEAK_GINST * 1e9 / per_scene, 0),
                                          "what": "all 26 batched kernels of a scene (EdgeConv 38 M, kNN 45.5 M, MLP1 12.8 M, moments, FPS, sort ...) as wave64 VALU instructions, "
                                                  "against what 1,024 SIMDs issue; MFMA instructions and memory time not counted: a floor on the time, not a model of it"}
        # ... and against HBM (the north star asks for the fraction of the HBM roofline): every batched kernel's measured HBM bytes per scene (PMC, corrected as the
        # guide prescribes and as reported) x the scenes/s of the timed region / 8 TB/s
        hb_c, hb_u = pmc.get("hbm_bytes_per_scene_launch", {}) or {}, pmc.get("hbm_bytes_uncorrected_per_scene_launch", {}) or {}
        if hb_c and base_l:
            tot_c = sum(v_ * rawk.get(k_, {}).get("launches", base_l) / base_l for k_, v_ in hb_c.items())
            tot_u = sum(v_ * rawk.get(k_, {}).get("launches", base_l) / base_l for k_, v_ in hb_u.items())
            roofline["whole_job_hbm"] = {"hbm_bytes_per_scene": int(tot_c), "hbm_bytes_per_scene_as_reported": int(tot_u), "peak_tb_per_s": (HBM_PEAK_GBS / 1000.0),
                                         "achieved_tb_per_s": round(tot_c * value / world / 1e12, 3), "frac": round(tot_c * value / world / 1e12 / (HBM_PEAK_GBS / 1000.0), 4),
                                         "frac_as_reported": round(tot_u * value / world / 1e12 / (HBM_PEAK_GBS / 1000.0), 4),
                                         "what": "2 x FETCH_SIZE + WRITE_SIZE of all batched kernels of a scene (solo batched PMC passes) x scenes/s: the job is bound by instruction issue "
                                                 "(whole_job_valu), not by HBM"}
        mb = (pmc.get("mfma_busy_share", {}) or {}).get(dom)
        if mb is not None:
            roofline["mfma_busy_share_pmc"] = {"value": mb, "source": f"profiles/{PROFILE_TAG}_pmc_kernels.json (SQ_VALU_MFMA_BUSY_CYCLES, solo batched)"}

        with_files = {}
        if not args.no_files:
            # file side (model.py:536-547), reported separately: the same step + the 14 label files of every
            # scene written by the native writer pool (sg_writer_*), flushed inside the timed region.  Twice: on a memory file system
            # (/dev/shm) and on the temp directory's (the GPU boxes mount an overlay on a disk there: the writer pool ALONE tops out at
            # ~1,250 scenes/s of .npy on it whatever its thread count, tools/time_writer.py -- that leg measures the file system)
            from seggroup_amd.model import AsyncLabelWriter
            writer = AsyncLabelWriter(threads=args.writer_threads)

            def fs_type(path):
                best, kind = "", "?"
                try:
                    for line in open("/proc/mounts"):
                        f = line.split()
                        if path.startswith(f[1]) and len(f[1]) > len(best):
                            best, kind = f[1], f[2]
                except OSError:
                    pass
                return kind

            def files_leg(base):
                out = {}
                for fmts in (("npy",), ("txt", "npy")):
                    with tempfile.TemporaryDirectory(prefix="sgbench_", dir=base) as td:
                        sub = scenes[:min(len(scenes), 64)]          # a bounded sample: txt is ~7 MB per scene
                        dirs = [os.path.join(td, sc_.name) for sc_ in sub]
                        # one untimed pass: creates the directories and the files' pages (later passes overwrite them, as a rerun of infer.py does)
                        runner.wait(runner.submit(sub, hip.MODE_INS_INFER, writer=writer, out_dirs=dirs, formats=fmts))
                        writer.flush()
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        reps_f, pend_f = 6, []
                        for _ in range(reps_f):                        # queued two ahead like the timed loop; the files of a pass overwrite the last
                            pend_f.append(runner.submit(sub, hip.MODE_INS_INFER, writer=writer, out_dirs=dirs, formats=fmts))
                            if len(pend_f) > 2:
                                runner.wait(pend_f.pop(0))
                        while pend_f:
                            runner.wait(pend_f.pop(0))
                        writer.flush()
                        out["+".join(fmts)] = round(reps_f * len(sub) / (time.perf_counter() - t1), 3)
                out["filesystem"] = "%s on %s" % (fs_type(base), base)
                return out

            shm = "/dev/shm"
            if os.path.isdir(shm) and os.access(shm, os.W_OK) and fs_type(shm) == "tmpfs":
                with_files = files_leg(shm)
                with_files["on_temp_dir"] = files_leg(tempfile.gettempdir())
            else:
                with_files = files_leg(tempfile.gettempdir())
            with_files["writer_threads"] = args.writer_threads
            writer.close()

        extras = {}
        if strong_scenes:
            # BASELINE configs[3] at W = 1: the 1201-scene set that `--scenes-total 1201` shards i mod W, ONE pass per step in batches of
            # --batch, everything resident; two passes must give the same labels (digest of the per-scene digests), and a strided sample is
            # checked against the single pipeline
            sb_batches = [strong_scenes[k:k + args.batch] for k in range(0, len(strong_scenes), args.batch)]
            run_batches(sb_batches[:2], 1)
            passes = []
            for _ in range(2):                                     # timed: results consumed (waited for, metric tensors read) like the main loop
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run_batches(sb_batches, 1, lambda res: [r_.acc for r_ in res])
                torch.cuda.synchronize()
                passes.append(time.perf_counter() - t1)
            dig = []
            for _ in range(2):                                     # untimed: sha256 over 8.4 MB of labels per scene is ~4 ms of host time each
                got = []
                run_batches(sb_batches, 1, lambda res: got.extend(label_digest(r_) for r_ in res))
                dig.append(got)
            idx = list(range(0, len(strong_scenes), 50))
            same_s = all(dig[0][i] == label_digest(solo.forward(strong_scenes[i], hip.MODE_INS_INFER)) for i in idx)
            h = hashlib.sha256("".join(dig[0]).encode()).hexdigest()
            extras["strong_1201"] = {"scenes": len(strong_scenes), "scenes_per_s": round(len(strong_scenes) / min(passes), 3),
                                     "ms_per_pass": [round(p_ * 1e3, 2) for p_ in passes], "passes_equal": dig[0] == dig[1],
                                     "parity_digest_ok": bool(same_s and dig[0] == dig[1]), "checked_against_single_pipeline": len(idx),
                                     "digest_of_digests": h,
                                     "what": "python bench.py --gpus 1 --scenes-total %d on the same engine: scene i of seeds 40000+i, the N = 1 point of the "
                                             "strong-scaling curve" % len(strong_scenes)}
            if not (same_s and dig[0] == dig[1]):
                parity_all = False
        if extras_on:
            # configs[1] / configs[4]: ONE scene alone on the GPU (sg_pipeline_forward, default stream): wall time per forward incl. the
            # host grouping and the D2H of the 14 label vectors
            lat = {}
            for tag, sc_, pl in (("150k", scenes[0], solo),) + ((("%dk" % (args.extra_stress // 1000), stress_scenes[0], None),) if stress_scenes else ()):
                own = pl is None
                if own:
                    pl = Pipeline(W, sc_.N, sc_.S, sc_.E0, sc_.V, stream=None, device=dev)
                    pl.set_timing(0)
                ts = []
                for it in range(7):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    pl.forward(sc_, hip.MODE_INS_INFER)
                    ts.append((time.perf_counter() - t1) * 1e3)
                lat[tag] = {"points": sc_.N, "segments": sc_.S, "ms_median": round(float(np.median(ts[2:])), 3), "ms_min": round(min(ts[2:]), 3)}
                if own:
                    pl.close()
            extras["latency_ms_single_scene"] = lat
        solo.close()
        if extra_scenes:
            # the same engine on ScanNet-shaped scenes (synthetic.make_scannet_scene: surfaces, 10k-30k-point floor / wall segments,
            # median segment ~55 points, V != N, every other scene with 15 % duplicated points); each scene checked against a
            # single pipeline like the main batch
            runner.run(extra_scenes, hip.MODE_INS_INFER)
            torch.cuda.synchronize()
            runner.reset_stage_stats()
            # the second headline: timed like the main loop -- regions of `passes` passes over the 64 scenes, queued two
            # ahead, back to back -- and reported like `repeat_values`
            passes, regions_x, res_x = max(4, args.steps // 4), [], None
            for _ in range(max(1, args.repeats)):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                pend = []
                for _p in range(passes):
                    pend.append(runner.submit(extra_scenes, hip.MODE_INS_INFER))
                    if len(pend) > 2:
                        res_x = runner.wait(pend.pop(0))
                while pend:
                    res_x = runner.wait(pend.pop(0))
                torch.cuda.synchronize()
                regions_x.append(round(passes * len(extra_scenes) / (time.perf_counter() - t1), 3))
            solo2 = Pipeline(W, *caps, stream=None, device=dev)
            solo2.set_timing(0)
            same_x = all(label_digest(res_x[i]) == label_digest(solo2.forward(extra_scenes[i], hip.MODE_INS_INFER)) for i in range(min(4, len(extra_scenes))))
            solo2.close()
            seg_max = max(int(s_.h_seg_size.max()) for s_ in extra_scenes)
            stage_x = {k_: round(v, 4) for k_, v in runner.mean_stage_ms().items() if v > 0}
            extras["scannet_profile"] = {"scenes_per_s": regions_x[0], "repeat_values": {"scenes_per_s": regions_x, "min": min(regions_x), "median": float(np.median(regions_x)),
                                                                                        "what": f"{len(regions_x)} timed regions of {passes} passes over {len(extra_scenes)} scenes each"},
                                         "ratio_to_value": round(float(np.median(regions_x)) / float(np.median(repeat_values)), 4),
                                         "scenes": len(extra_scenes), "largest_segment_points": seg_max,
                                         "equals_single_pipeline": bool(same_x), "stage_ms": stage_x,
                                         "cluster_trace_scene0": list(res_x[0].trace)}
            if not same_x:
                parity_all = False

        if world == 1 and not args.no_extras and args.extra_train > 0:
            # one training step (forward with tape + loss + backward + optimizer) per scene, batch 1 like train.py
            from seggroup_amd import train as _train, trainer as _trainer
            st = _train.initial_state(1)
            st.update({k_: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k_, v in weights.to_state_dict(W, prefix="").items()})
            caps_t = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
            trn = _trainer.Trainer(st, caps_t, device=dev)
            tt = {"forward": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}
            for it in range(-2, args.extra_train):
                sc_ = scenes[it % len(scenes)]
                torch.cuda.synchronize(); a0 = time.perf_counter()
                trn.forward(sc_)
                torch.cuda.synchronize(); a1 = time.perf_counter()
                mk = trn.dropout_mask("random")
                trn.loss(mk)
                torch.cuda.synchronize(); a2 = time.perf_counter()
                trn.backward(mk)
                torch.cuda.synchronize(); a3 = time.perf_counter()
                trn.average_gradients(); trn.optimizer_step()
                torch.cuda.synchronize(); a4 = time.perf_counter()
                if it >= 0:
                    tt["forward"] += a1 - a0; tt["loss"] += a2 - a1; tt["backward"] += a3 - a2; tt["optimizer"] += a4 - a3
            extras["train_step"] = {"ms_per_step": round(sum(tt.values()) / args.extra_train * 1e3, 3), "steps": args.extra_train,
                                    "ms": {k_: round(v / args.extra_train * 1e3, 3) for k_, v in tt.items()},
                                    "note": "one scene per step on one stream (train.py's batch size 1), SGD; off the headline metric"}
            trn.close()
            # ... and eight scenes per optimizer step (BatchTrainer: eight lanes on their own streams, gradients averaged like DDP ranks')
            lanes = min(8, len(scenes))
            if lanes > 1:
                btr = _trainer.BatchTrainer(st, caps_t, lanes=lanes, device=dev)
                tb, nb_ = 0.0, max(2, args.extra_train // 2)
                for it in range(-1, nb_):
                    grp = [scenes[(it * lanes + k_) % len(scenes)] for k_ in range(lanes)]
                    torch.cuda.synchronize(); a0 = time.perf_counter()
                    btr.step(grp)
                    torch.cuda.synchronize()
                    if it >= 0:
                        tb += time.perf_counter() - a0
                extras["train_step"]["batched"] = {"scenes_per_step": lanes, "ms_per_step": round(tb / nb_ * 1e3, 3),
                                                   "ms_per_scene": round(tb / nb_ / lanes * 1e3, 3), "steps": nb_}
                btr.close()

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_ref
            from threadpoolctl import threadpool_limits
            from seggroup_amd.numa import unbound
            # The rank is bound to its GPU's NUMA node (half of a two-socket host); the CPU leg gets the WHOLE host back for its duration
            # (every thread of the process, BLAS / OpenMP pools included: ADVICE round 4), and `cores` / the curve's "all cores" point are
            # what sched_getaffinity grants inside that window, not os.cpu_count().
            with unbound():
                host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
                # bounded thread count: the oracle is many small NumPy/torch ops, oversubscribing a 256-core host
                # makes it ~10x slower than 16 threads -- the curve below is measured here, on this box, and travels in the line
                used = min(host_cores, args.cpu_threads)
                times, same = [], True

                def one_run(nthreads):
                    torch.set_num_threads(nthreads)
                    with threadpool_limits(limits=nthreads):
                        t1 = time.perf_counter()
                        ref_ = cpu_ref.forward_scene(host_scene0, W, "ins_infer", faithful=True)
                        return time.perf_counter() - t1, ref_
                while len(times) < 2 and sum(times) < 40.0:                 # bounded: the CPU leg is ~35 s in total (two runs here + the curve below)
                    dt, ref = one_run(used)
                    times.append(dt)
                    same = same and ref["trace"] == batch_trace0 if args.scenes_total == 0 else same
                best, med = min(times), float(np.median(times))
                # one run each at neighbouring thread counts; --cpu-curve-full adds 64 threads and ALL granted cores (minutes on a 256-core
                # host: oversubscribed BLAS threads make that point the slowest by far, which is why the default run does not take it)
                curve = {str(used): round(best, 2)}
                for nt in sorted(({8, 32} | ({64, host_cores} if args.cpu_curve_full else set())) - {used}):
                    if nt > host_cores:
                        continue
                    dt, _ = one_run(nt)
                    curve[str(nt)] = round(dt, 2)
                torch.set_num_threads(used)
            anchor = None
            apath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_cpu_anchor_check.json")
            if not os.path.exists(apath):
                apath = os.path.join(ROOT, "profiles", "r04_cpu_anchor_check.json")
            if os.path.exists(apath):
                anchor = json.load(open(apath))
            cpu = {"value": round(1.0 / best, 5), "unit": "scenes/s", "cores": used, "host_cores": host_cores, "kind": "port",
                   "runs": len(times), "seconds_min_median": [round(best, 2), round(med, 2)],
                   "seconds_per_scene_by_threads": curve,
                   "affinity": f"CPU leg run with the process's pre-bind CPU mask restored ({host_cores} CPUs; the rank itself is bound to "
                               f"{numa_info.get('cpus_after')} for the GPU legs)",
                   "sample": f"1 scene of the same workload ({args.points} pts / {args.segments} segs), oracle/cpu_ref.py faithful mode, "
                             f"{len(times)} runs on {used} of {host_cores} host cores (value = best run; `seconds_per_scene_by_threads` = one run each at other "
                             f"thread counts: {used} is the fastest or close to it; 64 threads and all {host_cores} cores only with --cpu-curve-full); "
                             f"cluster trace equals the HIP path: {same}",
                   "anchor_check": anchor}

        if comm is None:
            comm = comm_probe(world, args.backend, dev)               # N = 1: behind every measured leg (see above)
        I_s, U_s = vec[:40], vec[40:80]
        I_i, U_i = vec[80:120], vec[120:160]
        with np.errstate(divide="ignore", invalid="ignore"):
            miou_sem = float(np.nanmean(I_s / U_s)) if np.any(U_s > 0) else float("nan")
            miou_ins = float(np.nanmean(I_i / U_i)) if np.any(U_i > 0) else float("nan")
        if args.scenes_total > 0:
            workload = (f"{args.scenes_total} distinct synthetic scenes sharded i mod {world} over {world} GPU(s), one pass per step in batches of "
                        f"{args.batch}, {args.points} pts / {args.segments} segs each (BASELINE.json configs[3])")
        else:
            workload = (f"{args.batch} distinct synthetic scenes per GPU per step, {args.points} pts / {args.segments} segs each "
                        f"(BASELINE.json configs[2]; configs[1] = the same scene shape, single scene)")
        out = {
            "metric": "pseudo-label scenes/sec (150k pts, 1.5k segs)", "value": round(value, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.scenes_total > 0 else "weak", "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": workload, "mode": "ins_infer", "scenes_per_step_per_gpu": len(scenes), "scenes_in_flight": args.groups * args.per_group,
                       "engine": f"{args.groups} groups x {args.per_group} scenes per batched launch",
                       "seg_profile": args.seg_profile, "weights": "tests/golden/weights_g2.npz", "parallelism": f"scene-parallel x{world}"},
            "timed_region_s": round(elapsed, 3),
            "repeat_values": {"scenes_per_s": repeat_values, "min": min(repeat_values), "median": float(np.median(repeat_values)),
                              "what": f"{len(repeat_values)} timed regions of {args.steps} steps each, back to back; `value` is the first"},
            "roofline": roofline, "cpu_baseline": cpu, "comm": comm,
            "parity_check": {"scenes_per_rank": n_par, "ranks_equal": parity_all,
                             "out_of_step": {"engine": "16 groups x 1 scene", "runs": oos_runs, "scene_results": oos_runs * n_par, "wrong_on_rank0": oos_wrong},
                             "what": "sha256 over the 14 label vectors + metric tensors + cluster trace of every checked scene of the last timed batch "
                                     "== the same scenes through one default-stream pipeline (and the extra legs' own checks)"},
            "with_label_files_scenes_per_s": with_files or None,
            "pseudo_label_mIoU": {"semantic": round(miou_sem, 4), "instance": round(miou_ins, 4), "scenes": int(vec[164])},
            "extra": extras or None,
            "engine_profile": engine_profile, "numa": numa_info,
            "cluster_trace_scene0": batch_trace0, "scene_generation_s": round(gen_s, 1),
        }
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        if not parity_all:
            print("bench.py: PARITY FAILURE -- the concurrent path's labels differ from the single-pipeline path", file=sys.stderr)
            rc = 4
    else:
        solo.close()
    runner.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
