#!/usr/bin/env python3
"""Headline benchmark: pseudo-label scenes/sec on synthetic ScanNet-shaped scenes (150k points /
1.5k segments), SegModel.forward in ins_infer mode through the C ABI (sg_pipeline_forward).

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

A "step" = one batch of `--batch` distinct scenes per GPU (weak scaling: per-GPU work is fixed as N
grows; scenes are independent, no collective in the data path -- one RCCL all-reduce of the metric
accumulators at the end).  Scenes are staged in HBM before the timed region; the timed
region covers everything SegModel.forward does for a scene (all kernels, the serial host grouping,
D2H of the 14 label vectors and metrics) except writing the label files (reported separately as
`with_label_files_scenes_per_s`, through the asynchronous native writer pool).  Several scenes are in flight per GPU (`--inflight` pipelines on separate HIP
streams, driven by native host threads inside sg_batch_forward) so the host's serial grouping phases overlap
other scenes' kernels.

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job scenes/s, plus
  roofline     - the dominant kernel stage measured with HIP events on the pipelines' own streams
                 inside the timed region (algorithmic bytes/flops per launch from DESIGN.md);
  cpu_baseline - the NumPy oracle ("port", faithful per-edge loops) timed on this box's host cores on a
                 bounded sample (one scene of the same workload), rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight pipeline (HIP's default of 4 makes 4 streams + the null stream share queues);
# must be set before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak


def kernel_model(n_points: int, k: int = 20):
    """Hot kernels: the pipeline stages that time them (HIP events), launches per scene and the ALGORITHMIC
    work of one launch."""
    n = float(n_points)
    c1, c12 = 2.0 * k * n * (18 * 64), 2.0 * k * n * (18 * 64 + 64 * 64)
    return {
        # name: (stages, launches per scene, bound, units per launch, unit, peak, scale)
        # S2X = MLP3's conv1' -> conv2 + BN2 statistics + max over k (its single evaluation); S1X = the same for MLP2's
        # one layer (the stage times include the ~20 us one-block BN fold).  MLP3's inner BN statistics come from
        # k_edge_moments (VALU; stage l3.edgeconv.stats1), which has no MFMA work to price.
        "k_edgeconv<S2X>": (["l3.edgeconv.stats2"], 1, "mfma", c12, "TFLOP/s", MFMA_F32_PEAK_TF, 1e12),
        "k_edgeconv<S1X>": (["l2.edgeconv.stats1"], 1, "mfma", c1, "TFLOP/s", MFMA_F32_PEAK_TF, 1e12),
        # kNN: reads [N,4] f32, writes [N,20] i32 (VALU-bound brute force inside clusters; HBM is its nominal roof)
        "k_cluster_knn_sorted": (["l2.knn", "l3.knn"], 2, "hbm", 96.0 * n, "GB/s", HBM_PEAK_GBS, 1e9),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="distinct scenes per GPU per step")
    ap.add_argument("--inflight", type=int, default=16, help="pipelines (HIP streams) per GPU")
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--segments", type=int, default=1500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline leg")
    ap.add_argument("--no-files", action="store_true", help="skip the separate with-files measurement")
    ap.add_argument("--writer-threads", type=int, default=16, help="native writer threads for the with-files leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU rehearsals)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from seggroup_amd import hip, synthetic, weights
    from seggroup_amd.scene import DeviceScene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    hip.require_device()
    local = local % torch.cuda.device_count()          # several ranks on one GPU only in gloo rehearsals
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
    # distinct synthetic scenes per rank (config 3 of BASELINE.json: batches of 150k/1.5k scenes)
    t0 = time.time()
    host_scenes = [synthetic.make_scene(args.points, args.segments, 30000 + 1000 * rank + i) for i in range(args.batch)]
    scenes = [DeviceScene.from_synthetic(s, device=dev) for s in host_scenes]
    gen_s = time.time() - t0
    from seggroup_amd.model import BatchRunner
    # timing level 1: HIP events only around the modelled kernels (every stage = ~25 events per scene = ~6 % of the throughput)
    runner = BatchRunner(W, scenes, inflight=args.inflight, device=dev, timing=1)
    acc = {"iou_sem": np.zeros(80), "iou_ins": np.zeros(80), "acc": np.zeros(4), "n": 0}

    def step(record=True):
        """One step = every scene of the batch through SegModel.forward (sg_batch_forward: native host threads)."""
        res = runner.run(scenes, hip.MODE_INS_INFER)
        if record:
            for r in res:
                acc["iou_sem"] += r.iou_sem.reshape(-1)
                acc["iou_ins"] += r.iou_ins.reshape(-1)
                acc["acc"] += np.nan_to_num(r.acc)
                acc["n"] += 1
        return [r.trace for r in res]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(record=False)
    runner.reset_stage_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        traces = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_scenes = world * args.batch * args.steps
    value = total_scenes / elapsed

    # the path's only collective: one all-reduce of the float64 metric accumulators
    vec = torch.from_numpy(np.concatenate([acc["iou_sem"], acc["iou_ins"], acc["acc"], [acc["n"]]]))
    if args.backend == "nccl":
        vec = vec.to(dev)
    if world > 1:
        dist.all_reduce(vec)
    vec = vec.cpu().numpy()

    out = None
    if rank == 0:
        mean_ms = runner.mean_stage_ms()
        model = kernel_model(args.points)
        per_scene = {kn: sum(mean_ms.get(st, 0.0) for st in m[0]) for kn, m in model.items()}
        dom = max(per_scene, key=per_scene.get)                 # kernel with the largest device time per scene
        stages, launches, bound, units, unit, peak, scale = model[dom]
        ms_launch = per_scene[dom] / launches
        achieved = units / (ms_launch * 1e-3) / scale
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and args.points == 150000:
            traffic = json.load(open(tpath)).get("bytes_per_launch", {}).get(dom)
        roofline = {"kernel": dom, "bound": bound, "achieved": round(achieved, 4), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 6), "traffic": traffic, "ms_per_launch": round(ms_launch, 4),
                    "launches_per_scene": launches, "measured_with": "HIP events on the pipeline streams, inside the timed region",
                    "kernel_ms_per_scene": {kn: round(v, 4) for kn, v in per_scene.items()},
                    # the same figures for every modelled kernel (the in-cluster kNN is VALU/latency-bound: its HBM
                    # fraction is nominal; the EdgeConv passes are the MFMA-bound ones)
                    "all_kernels": {kn: {"bound": m[2], "achieved": round(m[3] / (per_scene[kn] / m[1] * 1e-3) / m[6], 3), "peak": m[5],
                                         "unit": m[4], "frac": round(m[3] / (per_scene[kn] / m[1] * 1e-3) / m[6] / m[5], 5)}
                                    for kn, m in model.items() if per_scene[kn] > 0},
                    "stage_ms": {k_: round(v, 4) for k_, v in mean_ms.items()}}
        # the same kernels with ONE scene in flight (outside the timed region): with `inflight` streams sharing the GPU a
        # launch's duration says how long it shared the machine, not how well it uses it
        solo_runner = BatchRunner(W, scenes[:4], inflight=1, device=dev, timing=1)
        solo_runner.run(scenes[:2], hip.MODE_INS_INFER)
        solo_runner.reset_stage_stats()
        solo_runner.run(scenes[:4], hip.MODE_INS_INFER)
        solo_ms = solo_runner.mean_stage_ms()
        solo_runner.close()
        solo_scene = {kn: sum(solo_ms.get(st, 0.0) for st in m[0]) for kn, m in model.items()}
        roofline["single_stream"] = {kn: {"ms_per_launch": round(solo_scene[kn] / m[1], 4), "bound": m[2],
                                          "achieved": round(m[3] / (solo_scene[kn] / m[1] * 1e-3) / m[6], 3), "unit": m[4],
                                          "frac": round(m[3] / (solo_scene[kn] / m[1] * 1e-3) / m[6] / m[5], 5)}
                                     for kn, m in model.items() if solo_scene[kn] > 0}

        with_files = {}
        if not args.no_files:
            # file side (model.py:536-547), reported separately: the same step + the 14 label files of every
            # scene written by the native writer pool (sg_writer_*), flushed inside the timed region
            from seggroup_amd.model import AsyncLabelWriter
            writer = AsyncLabelWriter(threads=args.writer_threads)
            for fmts in (("npy",), ("txt", "npy")):
                with tempfile.TemporaryDirectory(prefix="sgbench_") as td:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    sub = scenes[:min(len(scenes), 4 * args.inflight)]          # a bounded sample: txt is ~7 MB per scene
                    dirs = [os.path.join(td, sc_.name) for sc_ in sub]
                    for _ in range(2):
                        runner.run(sub, hip.MODE_INS_INFER, writer=writer, out_dirs=dirs, formats=fmts)
                    writer.flush()
                    with_files["+".join(fmts)] = round(2 * len(sub) / (time.perf_counter() - t1), 3)
            writer.close()

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_ref
            from threadpoolctl import threadpool_limits
            # bounded thread count: the oracle is many small NumPy/torch ops, oversubscribing a 256-core host
            # makes it ~10x slower than 16 threads
            cores = min(os.cpu_count() or 1, args.cpu_threads)
            torch.set_num_threads(cores)
            t1 = time.perf_counter()
            with threadpool_limits(limits=cores):
                ref = cpu_ref.forward_scene(host_scenes[0], W, "ins_infer", faithful=True)
            cpu_s = time.perf_counter() - t1
            same = ref["trace"] == list(traces[0])
            cpu = {"value": round(1.0 / cpu_s, 5), "unit": "scenes/s", "cores": cores, "kind": "port",
                   "sample": f"1 scene of the same workload ({args.points} pts / {args.segments} segs), oracle/cpu_ref.py "
                             f"faithful mode, {cpu_s:.1f} s; cluster trace equals the HIP path: {same}"}

        I_s, U_s = vec[:40], vec[40:80]
        I_i, U_i = vec[80:120], vec[120:160]
        with np.errstate(divide="ignore", invalid="ignore"):
            miou_sem = float(np.nanmean(I_s / U_s)) if np.any(U_s > 0) else float("nan")
            miou_ins = float(np.nanmean(I_i / U_i)) if np.any(U_i > 0) else float("nan")
        out = {
            "metric": "pseudo-label scenes/sec (150k pts, 1.5k segs)", "value": round(value, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.batch} distinct synthetic scenes per GPU per step, {args.points} pts / {args.segments} "
                                   f"segs each (BASELINE.json configs[2]; configs[1] = the same scene shape, single scene)",
                       "mode": "ins_infer", "scenes_per_step_per_gpu": args.batch, "inflight_pipelines": args.inflight,
                       "weights": "tests/golden/weights_g2.npz", "parallelism": f"scene-parallel x{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
            "with_label_files_scenes_per_s": with_files or None,
            "pseudo_label_mIoU": {"semantic": round(miou_sem, 4), "instance": round(miou_ins, 4), "scenes": int(vec[164])},
            "cluster_trace_scene0": list(traces[0]), "scene_generation_s": round(gen_s, 1),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
