#!/usr/bin/env python3
"""Inference driver: the drop-in for the reference's `seggroup/infer.py` (pseudo-label generation).

Same command line (infer.py:195-212): `-n/--exp_name`, `--label_style`, `--sem_infer | --ins_infer`,
`-j/--workers`, `--no_cuda` (rejected: there is no CPU path), `--seed`, `-v/--visualize` (ignored);
same checkpoint (`checkpoints/<exp>/models/last.t7`, keys with or without `module.`), same
CWD-relative dataset tree, same `results/<exp>/<scene>/<mode>/*.txt` outputs (+ `.npy` twins), same
log lines in `checkpoints/<exp>/run_infer.log`.

Parallelism: one process per GPU; rank r handles scenes {i : i mod W == r} of the scene
list -- scenes are independent, so there is NO collective in the loop (the reference all-reduces three
tiny tensors per scene, infer.py:154-156, which lock-steps the ranks).  One all-reduce of the float64
accumulators [2*40 + 2*40 + 4 + 1] at the end; rank 0 prints the reference's summary.
`--sampler reference` reproduces DistributedSampler's shuffled, padded order (1201 -> 1208 at 8 GPUs)
so that the reduced metrics match the reference's double counting of the 7 padded scenes.

Launch:  python -m seggroup_amd.infer -n EXP --ins_infer            (spawns one process per visible GPU)
    or:  torchrun --nproc-per-node N -m seggroup_amd.infer -n EXP --ins_infer
"""
from __future__ import annotations

import os as _os
# one hardware queue per in-flight pipeline; must be set before the HIP runtime initialises (see bench.py)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import argparse
import os
import sys
import time
from typing import Callable, List, Optional

import numpy as np

SEM_VALID_CLASS_IDS = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
INS_VALID_CLASS_IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
SEM_CLASS_LABELS = ['wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture',
                    'counter', 'desk', 'curtain', 'refrigerator', 'shower curtain', 'toilet', 'sink', 'bathtub', 'otherfurniture']
INS_CLASS_LABELS = SEM_CLASS_LABELS[2:]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description='Pseudo Label Inference')
    p.add_argument('-n', '--exp_name', required=True, type=str, default=None, help='Name of the experiment to resume.')
    p.add_argument('--label_style', type=str, default='manual', help='Style of weak labels.')
    p.add_argument('--sem_infer', action='store_true', help='Infer pseudo labels for semantic segmentation.')
    p.add_argument('--ins_infer', action='store_true', help='Infer pseudo labels for instance segmentation.')
    p.add_argument('-j', '--workers', default=8, type=int, metavar='N', help='Number of data loading workers (default: 8).')
    p.add_argument('--no_cuda', action='store_true', help="Don't use CUDA (rejected: the hot path is GPU only).")
    p.add_argument('--seed', type=int, default=1, metavar='S', help='Random seed (default: 1)')
    p.add_argument('-v', '--visualize', action='store_true', help='Visualize results (ignored).')
    # additions of this build
    p.add_argument('--root', type=str, default='.', help='directory holding dataset/, checkpoints/, results/ (default: CWD)')
    p.add_argument('--sampler', choices=['shard', 'reference'], default='shard',
                   help="scene->rank assignment: 'shard' = i mod W (no padding); 'reference' = DistributedSampler order")
    p.add_argument('--out-format', type=str, default='txt,npy', help='comma list of txt,npy')
    p.add_argument('--world-size', type=int, default=0, help='processes to spawn (default: one per visible GPU)')
    p.add_argument('--backend', type=str, default='nccl', help='torch.distributed backend (nccl = RCCL on ROCm)')
    p.add_argument('--numa', type=str, default='auto', choices=['auto', 'off'],
                   help="'auto' = bind this rank's process (engine, writer and loader threads) to the CPUs of its GPU's NUMA node")
    p.add_argument('--label-transfer', type=str, default='tables', choices=['full', 'tables'],
                   help="fast path: 'tables' = only the [14,S] label tables cross PCIe and the writer workers look the 14 vectors up "
                        "(same files, byte for byte); 'full' = the vectors themselves are copied (8.4 MB per 150k-vertex scene)")
    p.add_argument('--port', type=int, default=2344, help='rendezvous port on 127.0.0.1 (reference: 2344)')
    p.add_argument('--batch', type=int, default=64, help='scenes per batch in the packed fast path (0 = the per-scene SegModel.forward loop)')
    p.add_argument('--inflight', type=int, default=32,
                   help='scenes in flight per GPU in the packed fast path (engine groups of 8).  80 -- what the bench, whose inputs are resident, is fastest near -- through most of '
                        'round 6; with packs to read and files to write beside it the driver is fastest at 32 (2,048 scenes on tmpfs, tools/sweep_driver.py, three boxes: '
                        '16 / 24 / 32 / 40 / 48 / 56 / 80 in flight = 1,600-1,660 / 1,750-1,850 / 1,860-2,080 / 1,850-1,990 / 1,760-1,950 / 1,610-1,810 / 1,400-1,610 scenes/s): a '
                        'smaller engine is created sooner (0.07 against 0.18 s) and leaves the loader its share of the GPU')
    p.add_argument('--no-cache', action='store_true', help='do not build / use packed scene files (dataset/scannet/cache/...)')
    p.add_argument('--synthetic', type=int, default=0, metavar='N',
                   help='write N synthetic ScanNet-shaped scenes in the reference\'s on-disk layout under --root (and a random-init '
                        'checkpoint checkpoints/<exp>/models/last.t7 if there is none), then run on them (SURVEY.md section 5, config row); '
                        'refuses a --root that already holds a scene list')
    p.add_argument('--synthetic-points', type=int, default=150000, help='points per synthetic scene (BASELINE.json: 150k)')
    p.add_argument('--synthetic-segments', type=int, default=1500, help='over-segments per synthetic scene (BASELINE.json: 1.5k)')
    p.add_argument('--synthetic-profile', type=str, default='voronoi', choices=['voronoi', 'scannet'], help='segment-size profile of the synthetic scenes')
    return p


def _synthetic_scene_job(job):
    points, segments, seed, profile, name = job
    from . import synthetic
    kw = {} if profile == 'voronoi' else {'seg_profile': profile}
    return synthetic.make_scene(points, segments, seed, name=name, **kw)


def write_synthetic_tree(args) -> int:
    """`--synthetic N`: the input tree infer.py reads (model.py:669-724, data.py:28-38) made of synthetic scenes, and a checkpoint to resume.
    Runs BEFORE anything touches the GPU (the generator pool is a spawn-context process pool: NumPy / SciPy only)."""
    import torch
    from . import synthetic, weights
    listing = os.path.join(args.root, 'dataset', 'scannet', 'scannetv2_train.txt')
    if os.path.exists(listing):
        print('--synthetic: %s exists; refusing to overwrite a dataset tree (choose an empty --root)' % listing)
        raise SystemExit(1)
    n = int(args.synthetic)
    jobs = [(args.synthetic_points, args.synthetic_segments, 60000 + i, args.synthetic_profile, 'scene%04d_00' % i) for i in range(n)]
    workers = max(1, min(int(args.workers), n, (os.cpu_count() or 1)))
    if workers > 1 and n >= 4:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context('spawn')) as pool:
            scenes = list(pool.map(_synthetic_scene_job, jobs))
    else:
        scenes = [_synthetic_scene_job(j) for j in jobs]
    synthetic.write_reference_tree(args.root, scenes, label_style=args.label_style)
    ck = os.path.join(args.root, 'checkpoints', args.exp_name, 'models')
    os.makedirs(ck, exist_ok=True)
    if not os.path.exists(os.path.join(ck, 'last.t7')):
        torch.save({'state_dict': weights.to_full_state_dict(weights.make_weights(args.seed, bn1_gamma=2.0))}, os.path.join(ck, 'last.t7'))
    return n


from .util import IOStream  # noqa: E402,F401  (the reference keeps it in util.py)


def scene_indices(num_scenes: int, rank: int, world: int, sampler: str) -> List[int]:
    if sampler == 'shard':
        return list(range(rank, num_scenes, world))
    # DistributedSampler(shuffle=True, seed=0, epoch=0): randperm, pad by wrapping, stride by world (infer.py:98,136)
    import torch
    g = torch.Generator()
    g.manual_seed(0)
    idx = torch.randperm(num_scenes, generator=g).tolist()
    total = -(-num_scenes // world) * world
    idx += idx[:total - num_scenes]
    return idx[rank:total:world]


class Accumulator:
    """float64 sums of I/U per class and of the four accuracies (infer.py:140-169)."""

    def __init__(self):
        self.v = np.zeros(165, dtype=np.float64)

    def add(self, iou_sem, iou_ins, acc):
        self.v[0:80] += np.asarray(iou_sem, dtype=np.float64).reshape(-1)
        self.v[80:160] += np.asarray(iou_ins, dtype=np.float64).reshape(-1)
        self.v[160:164] += np.asarray(acc, dtype=np.float64)
        self.v[164] += 1

    def summary(self):
        I_s, U_s, I_i, U_i = self.v[0:40], self.v[40:80], self.v[80:120], self.v[120:160]
        n = max(self.v[164], 1)
        with np.errstate(divide='ignore', invalid='ignore'):
            return dict(iou_sem=I_s / U_s, iou_ins=I_i / U_i, acc_sem=self.v[160] / n, acc_ins=self.v[161] / n,
                        acc_sem_sel=self.v[162] / n, acc_ins_sel=self.v[163] / n, n=int(self.v[164]))


def progress_line(done, total, s) -> str:
    with np.errstate(invalid='ignore'):
        return ('Infer(%04d/%04d)    Instance mIoU: %.2f%%    Semantic mIoU: %.2f%%    Instance Acc: %.2f%%    Semantic Acc: %.2f%%'
                % (done, total, np.nanmean(s['iou_ins']) * 100, np.nanmean(s['iou_sem']) * 100, s['acc_ins'] * 100, s['acc_sem'] * 100))


def final_report(s, io: IOStream) -> None:
    """The '==> Infer' line and the per-class tables (infer.py:63-76,178-190)."""
    with np.errstate(invalid='ignore'):
        io.cprint('==> Infer           Instance mIoU: %.2f%%    Semantic mIoU: %.2f%%    Instance Acc: %.2f%%    Semantic Acc: %.2f%%'
                  % (np.nanmean(s['iou_ins']) * 100, np.nanmean(s['iou_sem']) * 100, s['acc_ins'] * 100, s['acc_sem'] * 100))
        sem_sel = s['iou_sem'][SEM_VALID_CLASS_IDS - 1]
        ins_sel = s['iou_ins'][INS_VALID_CLASS_IDS - 1]
        io.cprint('')
        io.cprint('Instance mIoU (18 classes): %.2f%%      Acc (18 classes): %.2f%%' % (np.nanmean(ins_sel) * 100, s['acc_ins_sel'] * 100))
        for i in range(18):
            io.cprint('{:<16}{:<16}'.format(INS_CLASS_LABELS[i], '%.2f%%' % (ins_sel[i] * 100)))
        io.cprint('')
        io.cprint('Semantic mIoU (20 classes): %.2f%%      Acc (20 classes): %.2f%%' % (np.nanmean(sem_sel) * 100, s['acc_sem_sel'] * 100))
        for i in range(20):
            io.cprint('{:<16}{:<16}'.format(SEM_CLASS_LABELS[i], '%.2f%%' % (sem_sel[i] * 100)))
        io.cprint('')


def load_checkpoint(model, ckpt_path: str) -> None:
    """`infer.py:112-121`: `last.t7` = {'state_dict': ...} saved from the DDP-wrapped model ('module.' prefix).  Loaded
    STRICTLY like the reference: a checkpoint of another experiment / with renamed or missing keys must not end up
    writing pseudo labels from randomly initialised weights."""
    import torch
    ckpt = torch.load(ckpt_path, map_location='cpu')
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)


def run_worker(rank: int, world: int, args, forward_fn: Optional[Callable] = None, init_dist: bool = True) -> Optional[dict]:
    """One rank's loop.  `forward_fn(scene_index) -> (iou_sem, iou_ins, acc)` replaces the model in the
    CPU (gloo) tests of the driver logic; by default it is SegModel.forward on this rank's GPU."""
    import torch
    import torch.distributed as dist

    io = IOStream(os.path.join(args.root, 'checkpoints', args.exp_name, 'run_infer.log')) if rank == 0 else None
    if world > 1 and init_dist and not dist.is_initialized():
        dist.init_process_group(backend=args.backend, init_method=f'tcp://127.0.0.1:{args.port}', world_size=world, rank=rank)
    with open(os.path.join(args.root, 'dataset', 'scannet', 'scannetv2_train.txt')) as f:
        scene_list = f.readlines()

    dev = None
    if forward_fn is None:
        # FIRST: this process onto the CPUs of its GPU's NUMA node -- every thread it has by now (HIP runtime's included) and all later ones, so that
        # everything it first-touches from here on lies on that node: the engine's pinned buffers, the loader's staging buffers AND the packs a first run
        # writes (page cache / tmpfs pages stay where their writer ran).  Round 6, 2,048 scenes on tmpfs, packs written from the GPU's node against
        # the other one: 1,860-1,955 against 1,513-1,668 scenes/s in the runs that read them -- the bind used to come behind the pack build
        # (tools/sweep_driver.py --pre-bind)
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
        dev = torch.device('cuda', torch.cuda.current_device())
        from .numa import bind_to_gpu_node
        numa = bind_to_gpu_node(dev.index, getattr(args, 'numa', 'auto'))
        if numa['bound']:
            print('[rank %d] GPU %s on NUMA node %d: bound to %d of %d CPUs' % (rank, numa['pci'], numa['numa_node'], numa['cpus_after'], numa['cpus_before']), flush=True)
    if forward_fn is None and args.batch > 0 and not args.no_cache:
        # packed fast path: this rank's missing scene packs are built first (native threads; the Python fallback runs in threads of this process too)
        from . import cache
        mine_names = [scene_list[i][:-1] for i in scene_indices(len(scene_list), rank, world, args.sampler)]
        # the reference gives every rank workers / ngpus loader processes (infer.py:94)
        per_rank = max(1, -(-int(args.workers) // max(world, 1)))
        # ... the ONE-OFF pack build of a tree seen for the first time takes the rank's share of the host instead: a pack is ~20 ms of file
        # reads, a JSON parse and a 13 MB write that release the GIL (round 6: 260 -> see profiles/r06_driver_end_to_end.json `packed_cold`)
        # (threads: the native builder peaks between 16 and 32 -- 1 / 8 / 16 / 32 / 64 threads 182 / 1,123 / 2,054 / 1,791-2,069 / 1,333 packs/s of 150k-point scenes on tmpfs,
        # tools/time_pack_build.py, since its buffers are the threads' own)
        cold = max(per_rank, min(24, (len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)) // max(world, 1)))
        # ... in a thread beside the model's construction and the checkpoint's load below (the builder's threads are native, the call releases the GIL): the
        # ~0.25 s those take are not added to a first run's pack build
        import threading
        from . import hip
        hip.lib()                                       # loaded once, here, not by two threads at the same time
        build_box = {}

        def _build():
            try:
                t_b = time.time()
                build_box['built'] = cache.build_missing(args.root, sorted(set(mine_names)), args.label_style, workers=cold)
                build_box['s'] = time.time() - t_b
            except BaseException as e:                  # re-raised in the caller's thread below
                build_box['error'] = e
        build_thread = threading.Thread(target=_build, name='sg-pack-build')
        build_thread.start()
    else:
        build_thread = None
    if forward_fn is None:
        from .data import ScanNet
        from .model import SegModel
        model = SegModel(exp_name=args.exp_name, cuda=True, visualize=False, sem_infer=args.sem_infer, ins_infer=args.ins_infer,
                         data_root=args.root, out_formats=tuple(args.out_format.split(',')), label_style=args.label_style).to(dev)
        if rank == 0:
            io.cprint('Network parameters: {}'.format(sum(x.nelement() for x in model.parameters())))
        ckpt_path = os.path.join(args.root, 'checkpoints', args.exp_name, 'models', 'last.t7')
        if not os.path.exists(ckpt_path):
            if rank == 0:
                io.cprint('No checkpoint model, please make sure that you use right name in --exp_name')
            raise SystemExit(1)
        load_checkpoint(model, ckpt_path)
        if rank == 0:
            io.cprint('Load model from ' + ckpt_path)
        model.epoch = 'sem_infer' if args.sem_infer else 'ins_infer'
        dataset = ScanNet(label_style=args.label_style, root=args.root)
        if build_thread is not None:
            build_thread.join()
            if 'error' in build_box:
                raise build_box['error']
            if os.environ.get('SG_DRIVER_PROFILE'):
                print('[driver profile] pack build: %d packs on %d threads in %.3f s' % (build_box.get('built', 0), cold, build_box.get('s', 0.0)), flush=True)
            if build_box.get('built') and rank == 0:
                io.cprint('Built %d scene packs under dataset/scannet/cache/%s' % (build_box['built'], args.label_style))

        def forward_fn(i):   # noqa: F811
            data, weak, info = dataset[i]
            with torch.no_grad():
                out = model(data[None].to(dev, non_blocking=True), weak[None].to(dev, non_blocking=True), info[None])
            return tuple(o.cpu().numpy() for o in out)
        forward_fn.flush = model.flush

    mine = scene_indices(len(scene_list), rank, world, args.sampler)
    acc = Accumulator()
    t0 = time.time()
    fast = dev is not None and args.batch > 0
    if fast:
        # fast path: loader threads stage the next batch -- from scene packs (built once) or, with
        # --no-cache, straight from the reference's files (native seg.json parser, no pack written) -- while the GPU runs
        # the current one through the scene engine; label files are written by the native writer pool
        _run_packed(rank, world, args, model, scene_list, mine, acc, io, dev)
    else:
        for step, i in enumerate(mine):
            acc.add(*forward_fn(i))
            if rank == 0:   # rank 0's running view (the reference prints the all-reduced view every step)
                io.cprint(progress_line(min((step + 1) * world, len(scene_list)), len(scene_list), acc.summary()))
    if forward_fn is not None and hasattr(forward_fn, "flush"):
        forward_fn.flush()               # label files are written asynchronously: wait for them before reporting
    vec = torch.from_numpy(acc.v.copy())
    # a process group that exists is used, also at world size 1 (the RCCL communicator of a one-GPU run is built and exercised: the
    # `-m gpu` tests drive exactly this on a one-GPU box)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        if dev is not None and args.backend == 'nccl':
            vec = vec.to(dev)
        dist.all_reduce(vec)           # the ONLY collective: 165 float64 over RCCL/xGMI (1.3 KB, latency-bound)
        vec = vec.cpu()
    result = None
    if rank == 0:
        total = Accumulator()
        total.v = vec.numpy().copy()
        result = total.summary()
        result['elapsed_s'] = time.time() - t0
        if getattr(acc, 'startup_s', None) is not None:
            result['startup_s'], result['first_batch'] = acc.startup_s, acc.first_batch
        final_report(result, io)
        io.close()
    if world > 1 and init_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


def _run_packed(rank, world, args, model, scene_list, mine, acc, io, dev):
    from concurrent.futures import ThreadPoolExecutor
    from . import cache, hip
    from .model import AsyncLabelWriter, BatchRunner, Engine

    names = [scene_list[i][:-1] for i in mine]
    formats = tuple(args.out_format.split(','))
    mode = hip.MODE_SEM_INFER if args.sem_infer else hip.MODE_INS_INFER
    workers = max(1, int(args.workers))

    # Scene packs come through the native loader (csrc/loader.cpp): worker threads read them into pinned buffers and upload them into
    # device slots allocated here, once.  --no-cache
    # stages straight from the reference's files in Python threads (no pack written).
    loader, all_caps = None, None
    if not args.no_cache:
        paths = {n: cache.pack_scene(args.root, n, args.label_style) for n in dict.fromkeys(names)}
        slot_bytes = max((os.path.getsize(p) for p in paths.values()), default=1 << 20)
        # four batches' worth of slots: one in the engine, one queued behind it, two being loaded (two batches of requests are outstanding:
        # with one, the loop alternated between waiting for the loader and waiting for the engine)
        # eight loader threads read and upload ~2,800 packs/s (tools/time_loader.py: PCIe-bound at 45 GB/s from 16 on); more only take
        # cores from the writer pool
        # the slots' size: the largest pack + the largest adjacency widened to int64 (every pack's header says how many edges it holds:
        # ~20 us per file; sized by the worst case of 3 x the file, 256 slots of 150k-point packs were ~9 GB of device memory)
        dims = [cache.pack_dims(p) for p in paths.values()]
        max_edges = max((d['E0'] for d in dims), default=0)
        # ... and how large an engine this rank's scenes need: it is created below WHILE the loader reads the first batches (it used to be
        # sized by the first batch once that had arrived: ~0.1 s of 124 MB device slots allocated with nothing beside them)
        all_caps = tuple(max(d[k] for d in dims) for k in ('N', 'S', 'E0', 'V')) if dims else None
        # Round 6 sweep (tools/sweep_driver.py, 2,048 scenes on tmpfs, two boxes): 4 loader threads 1,895-1,930 scenes/s steady, 6 (the old default)
        # 1,824-1,852, 8 -- on a slower box -- 1,092 against 1,610-1,694 for 4-6: every further thread is another pack read + bulk upload competing
        # with the engine's own transfers; with `.txt` output 3-4 threads 1,478-1,488 against 1,445-1,466 for 8
        # ... and with 32 scenes in flight (the default since), 3 against 4 threads, A / B on two boxes: 1,825-1,893 against 1,517-1,831 scenes/s overall on one,
        # 1,443-1,673 against 1,734-1,861 on the other (whose host reads a pack more slowly): 4 stays.  Reading the next pack WHILE the last one uploads (two staging
        # buffers per worker) was built and measured too: the loader then idles half of the time and the run is no faster (3 / 4 threads 1,767-1,857 / 1,509-1,820
        # against 1,825-1,893 / 1,517-1,831 without) -- what the loader takes, the engine loses
        n_load = int(os.environ.get('SG_LOADER_THREADS', '0')) or min(4, workers)
        loader = cache.PackLoader(threads=n_load, slots=int(os.environ.get('SG_LOADER_SLOT_BATCHES', '4')) * max(args.batch, 1), slot_bytes=slot_bytes, device=dev,
                                  max_edges=max_edges)

        class _Loaded:                                      # a future-like handle on a loader ticket
            def __init__(self, t):
                self.t = t

            def result(self):
                return loader.wait(self.t)

    def stage(name):
        from .scene import DeviceScene
        return DeviceScene.from_reference_tree(name, root=args.root, label_style=args.label_style, device=dev)

    pool = ThreadPoolExecutor(max_workers=workers)

    def request(name):
        return _Loaded(loader.submit(paths[name])) if loader is not None else pool.submit(stage, name)

    batches = [names[k:k + args.batch] for k in range(0, len(names), args.batch)]
    ahead = int(os.environ.get('SG_DRIVER_AHEAD', '2')) if loader is not None else 1                 # batches of staging requests outstanding
    pending_q = [[request(n) for n in batches[k]] for k in range(min(ahead, len(batches)))]
    # Writer threads beyond what the formats need take memory bandwidth from the loader's copies (2,048 scenes on tmpfs, 256-core host,
    # tools/time_driver.py --out-format "npy@6;npy@16;txt,npy@8;txt,npy@16;txt,npy@32"): `.npy` only 1,430-1,500 scenes/s with 6 threads, 1,230
    # with 16; `.txt` + `.npy` (the text of a scene is ~3 ms of one core since the formatter takes small values from a table) 1,290 with 8,
    # 1,205 with 16, 1,150 with 32
    # Round 6, with 32 scenes in flight (same sweeps): `.npy` 6 / 8 writer threads 1,860-1,910 / 1,920-1,950 scenes/s; `.txt` + `.npy` 8 / 12 / 16 threads 1,470-1,510 /
    # 1,650-1,670 / 1,520-1,600.  The count no longer follows -j (the reference's loader processes): it is what the formats need, inside the rank's share of the host
    cpus = (len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)) // max(world, 1)
    writer = AsyncLabelWriter(threads=int(os.environ.get('SG_WRITER_THREADS', '0')) or max(2, min(12 if 'txt' in formats else 8, max(cpus - 6, 2))))
    runner, done, stalled = None, 0, []
    w = model.export_weights()
    tickets = []
    t_start, startup = time.time(), None
    if all_caps is not None and all(c > 0 for c in (all_caps[0], all_caps[1], all_caps[3])):
        groups, per_group = BatchRunner.shape(args.inflight)
        runner = Engine(w, all_caps, groups=groups, per_group=per_group, device=dev, timing=0, label_transfer=args.label_transfer)
    t_engine = time.time() - t_start

    prof = {"load_wait": 0.0, "submit": 0.0, "engine_wait": 0.0, "log": 0.0} if os.environ.get("SG_DRIVER_PROFILE") else None

    def consume(t):
        nonlocal done
        t_a = time.time()
        results = runner.wait(t)
        if prof is not None:
            prof["engine_wait"] += time.time() - t_a
            t_a = time.time()
        for s_, r in zip(t.scenes, results):
            acc.add(r.iou_sem, r.iou_ins, r.acc)
            done += 1
            if r.stalled:
                # the reference never returns from such a scene (pass 2 of group_nearby_clusters loops forever, model.py:
                # 228-239); here the sweep that made no progress was the last one -- say so, the labels are still written
                stalled.append(s_.name)
                print('[rank %d] %s: a <5-point cluster could not be merged (reference would not terminate); sweep cut short' % (rank, s_.name), flush=True)
            if rank == 0:
                io.cprint(progress_line(min(done * world, len(scene_list)), len(scene_list), acc.summary()))
            if hasattr(s_, "release"):
                s_.release()                                # the loader's slot is free for the batch after next
        if prof is not None:
            prof["log"] += time.time() - t_a

    for bi, batch in enumerate(batches):
        t_a = time.time()
        scenes = [f.result() for f in pending_q.pop(0)]
        if prof is not None:
            prof["load_wait"] += time.time() - t_a
        if bi + ahead < len(batches):
            pending_q.append([request(n) for n in batches[bi + ahead]])
        if runner is None or any(not runner.fits(s_) for s_ in scenes):
            while tickets:                                  # the engine is rebuilt with larger capacities: drain it first
                consume(tickets.pop(0))
            caps = runner.caps if runner is not None else None
            if runner is not None:
                runner.close()
            runner = BatchRunner(w, scenes, inflight=args.inflight, device=dev, min_caps=caps, timing=0, label_transfer=args.label_transfer)
        # one batch is queued behind the one in flight: the engine's groups never drain between batches
        t_a = time.time()
        tickets.append(runner.submit(scenes, mode, writer=writer, out_dirs=[model.output_root(s_.name) for s_ in scenes], formats=formats))
        if prof is not None:
            prof["submit"] += time.time() - t_a
        if startup is None:
            # one-off: the first batch staged with nothing to overlap it, the engine's slots (124 MB of device memory each) and the pinned
            # label ring created -- about a second that a short run cannot amortise
            startup = time.time() - t_start
            acc.startup_s, acc.first_batch = startup, len(scenes)
        if len(tickets) > 1:
            consume(tickets.pop(0))
    while tickets:
        consume(tickets.pop(0))
    t_a = time.time()
    writer.flush()
    if prof is not None:
        prof["final_flush"] = time.time() - t_a
        print("[driver profile] %s engine created in %.3f s, start-up %.3f s, total %.3f s" % ({k: round(v, 3) for k, v in prof.items()}, t_engine, startup or 0.0,
                                                                                              time.time() - t_start), flush=True)
    writer.close()
    if runner is not None:
        runner.close()
    if loader is not None:
        loader.close()
    pool.shutdown()
    if stalled and rank == 0:
        io.cprint('%d scene(s) hit the non-terminating pass-2 case of group_nearby_clusters: %s' % (len(stalled), ' '.join(stalled)))


def _spawn_entry(rank, world, args):
    run_worker(rank, world, args)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.sem_infer == args.ins_infer:
        print("Please choose either '--sem_infer' or '--ins_infer'")       # infer.py:214-216
        raise SystemExit(1)
    if args.synthetic > 0:                                                 # before the first HIP call: the generator pool spawns processes
        if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
            print('--synthetic writes the tree from ONE process: run it once without torchrun, then start the ranks on that --root')
            raise SystemExit(1)
        n_syn = write_synthetic_tree(args)                                 # refuses an existing tree before it creates anything
        os.makedirs(os.path.join(args.root, 'checkpoints', args.exp_name), exist_ok=True)
    import torch
    if args.no_cuda or not torch.cuda.is_available():
        print('seggroup_amd runs on MI355X only: no CPU fallback (use oracle/cpu_ref.py for testing)')
        raise SystemExit(1)
    np.seterr(divide='ignore', invalid='ignore')
    if int(os.environ.get('RANK', '0')) == 0:                              # once, like the reference (under torchrun every rank runs main())
        io = IOStream(os.path.join(args.root, 'checkpoints', args.exp_name, 'run_infer.log'))
        io.cprint(str(args))
        io.cprint("Let's use " + str(torch.cuda.device_count()) + " GPUs!")
        if args.synthetic > 0:
            io.cprint('Wrote %d synthetic scenes (%d points / %d segments, %s) under %s' % (n_syn, args.synthetic_points, args.synthetic_segments,
                                                                                         args.synthetic_profile, os.path.join(args.root, 'dataset', 'scannet')))
        io.close()
    torch.manual_seed(args.seed)
    np.random.seed(1)
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:              # launched by torchrun
        import torch.distributed as dist
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        if world > 1:
            dist.init_process_group(backend=args.backend)
        run_worker(rank, world, args, init_dist=False)
        if world > 1:
            dist.destroy_process_group()
        return
    world = args.world_size or torch.cuda.device_count()
    if world == 1:
        run_worker(0, 1, args)
    else:
        import torch.multiprocessing as mp
        mp.spawn(_spawn_entry, nprocs=world, args=(world, args))


if __name__ == '__main__':
    main()
