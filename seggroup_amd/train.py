#!/usr/bin/env python3
"""Training driver: the drop-in for the reference's `seggroup/train.py`.

Same command line (train.py:225-247): `-n/--exp_name`, `-r/--resume`, `--epochs` (6), `--label_style`, `--use_sgd` (True: SGD with
lr * 100, else Adam), `-j/--workers`, `--lr` (0.001), `--momentum` (0.9), `--no_cuda` (rejected: there is no CPU path), `--seed`,
`-v/--visualize` (ignored); same dataset tree, same pseudo-label files under `results/<exp>/<scene>/epoch_<n>/` (`epoch_last` for
the final epoch, model.py:688-691), same log lines in `checkpoints/<exp>/run.log`, same checkpoints
`checkpoints/<exp>/models/{epoch_<n>,last}.t7` = {'epoch', 'state_dict' (DDP's 'module.' keys), 'optimizer' (torch.optim layout)}, so
the reference's infer.py / --resume read what this writes and vice versa.

One process per GPU (train.py:255-257), batch size 1 per rank, scenes dealt by DistributedSampler's shuffled order per epoch.  Per
step and rank: forward + loss + backward on HIP (csrc/trainer.cpp), then ONE all-reduce over RCCL of the flat gradient vector
(0.59 MB) with the step's log terms riding behind it (the reference: DDP's bucketed gradient all-reduce + four more all-reduces
for loss / IoU / accuracy, train.py:170-173), then the optimizer kernel on every rank.

Launch:  python -m seggroup_amd.train -n EXP            (spawns one process per visible GPU)
    or:  torchrun --nproc-per-node N -m seggroup_amd.train -n EXP
"""
from __future__ import annotations

import argparse
import os
import time
from typing import Callable, Dict, List, Optional

import numpy as np

from .infer import IOStream, SEM_VALID_CLASS_IDS, INS_VALID_CLASS_IDS, SEM_CLASS_LABELS, INS_CLASS_LABELS


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description='Point-Level Pseudo Label Generation')
    p.add_argument('-n', '--exp_name', type=str, default=None, help='Name of the experiment (default is to use date_time).')
    p.add_argument('-r', '--resume', action='store_true', help='Resume training from the last checkpoint.')
    p.add_argument('--epochs', type=int, default=6, help='Number of the episode to train.')
    p.add_argument('--label_style', type=str, default='manual', help='Style of weak labels.')
    p.add_argument('--use_sgd', type=lambda s: str(s).lower() not in ('false', '0', 'no', ''), default=True, help='Use SGD.')
    p.add_argument('-j', '--workers', default=8, type=int, metavar='N', help='Number of data loading workers (default: 8).')
    p.add_argument('--lr', type=float, default=0.001, metavar='LR', help='Learning rate (default: 0.001, 0.1 if using sgd)')
    p.add_argument('--momentum', type=float, default=0.9, metavar='M', help='SGD momentum (default: 0.9)')
    p.add_argument('--no_cuda', action='store_true', help="Don't use CUDA (rejected: the hot path is GPU only).")
    p.add_argument('--seed', type=int, default=1, metavar='S', help='Random seed (default: 1)')
    p.add_argument('-v', '--visualize', action='store_true', help='Visualize results (ignored).')
    # additions of this build
    p.add_argument('--root', type=str, default='.', help='directory holding dataset/, checkpoints/, results/ (default: CWD)')
    p.add_argument('--out-format', type=str, default='txt,npy', help='comma list of txt,npy; empty = no pseudo-label files while training')
    p.add_argument('--world-size', type=int, default=0, help='processes to spawn (default: one per visible GPU)')
    p.add_argument('--backend', type=str, default='nccl', help='torch.distributed backend (nccl = RCCL on ROCm)')
    p.add_argument('--port', type=int, default=23456, help='rendezvous port on 127.0.0.1 (reference: 23456)')
    p.add_argument('--max-steps', type=int, default=0, help='stop every epoch after this many optimizer steps per rank (0 = the whole scene list)')
    p.add_argument('--scenes-per-step', type=int, default=1, dest='scenes_per_step',
                   help='scenes per optimizer step and GPU: > 1 runs that many forward / backward passes side by side (BatchTrainer) and averages their '
                        'gradients, as the reference does over ranks (not in the reference: its batch size per rank is 1)')
    p.add_argument('--no-cache', action='store_true', help='do not build / use packed scene files (dataset/scannet/cache/...)')
    p.add_argument('--param-digests', action='store_true', help='every rank writes sha256 of its final parameter vector to '
                   'checkpoints/<exp>/models/rank<r>.sha256 (the ranks must agree: same averaged gradient, same optimizer)')
    return p


def epoch_indices(num_scenes: int, rank: int, world: int, epoch: int) -> List[int]:
    """DistributedSampler(train_dataset) with set_epoch(epoch) (train.py:102,139): shuffle with seed 0 + epoch, pad by wrapping, stride"""
    import torch
    g = torch.Generator()
    g.manual_seed(0 + epoch)
    idx = torch.randperm(num_scenes, generator=g).tolist()
    total = -(-num_scenes // world) * world
    idx += idx[:total - num_scenes]
    return idx[rank:total:world]


class EpochLog:
    """rank 0's running sums of train.py:141-150,175-189 (every term already summed over the ranks of a step)"""

    def __init__(self):
        self.v = np.zeros(166, dtype=np.float64)

    def add(self, summed: np.ndarray) -> None:
        self.v += np.asarray(summed, dtype=np.float64)[:166]

    def line(self, head: str) -> str:
        v = self.v
        n = max(v[165], 1.0)                                   # scenes seen so far over all ranks = (i + 1) * ngpus
        with np.errstate(divide='ignore', invalid='ignore'):
            iou_sem, iou_ins = v[1:41] / v[41:81], v[81:121] / v[121:161]
            return head + '    Loss: %.6f    Instance mIoU: %.2f%%    Semantic mIoU: %.2f%%    Instance Acc: %.2f%%    Semantic Acc: %.2f%%' % (
                v[0] / n, np.nanmean(iou_ins) * 100, np.nanmean(iou_sem) * 100, v[162] / n * 100, v[161] / n * 100)

    def class_report(self, io: IOStream) -> None:
        v = self.v
        n = max(v[165], 1.0)
        with np.errstate(divide='ignore', invalid='ignore'):
            sem_sel, ins_sel = (v[1:41] / v[41:81])[SEM_VALID_CLASS_IDS - 1], (v[81:121] / v[121:161])[INS_VALID_CLASS_IDS - 1]
            io.cprint('')
            io.cprint('Instance mIoU (18 classes): %.2f%%      Acc (18 classes): %.2f%%' % (np.nanmean(ins_sel) * 100, v[164] / n * 100))
            for i in range(18):
                io.cprint('{:<16}{:<16}'.format(INS_CLASS_LABELS[i], '%.2f%%' % (ins_sel[i] * 100)))
            io.cprint('')
            io.cprint('Semantic mIoU (20 classes): %.2f%%      Acc (20 classes): %.2f%%' % (np.nanmean(sem_sel) * 100, v[163] / n * 100))
            for i in range(20):
                io.cprint('{:<16}{:<16}'.format(SEM_CLASS_LABELS[i], '%.2f%%' % (sem_sel[i] * 100)))
            io.cprint('')


# ---- checkpoints in the reference's layout (train.py:217-221) -----------------------------------------------------------------
def optimizer_state_dict(tr) -> Dict[str, object]:
    """torch.optim.SGD / Adam `state_dict()` layout over the 19 parameter tensors in named_parameters() order, so that the
    reference's `optimizer.load_state_dict(checkpoint['optimizer'])` (train.py:128) accepts it"""
    import torch
    from .trainer import param_slots, PARAM_SHAPES
    st = tr.optimizer_state()
    state = {}
    for i, (name, off, cnt) in enumerate(param_slots()):
        shape = PARAM_SHAPES.get(name, (cnt,))
        if st["steps"] == 0:
            continue
        if st["kind"] == "sgd":
            state[i] = {"momentum_buffer": st["a"][off:off + cnt].clone().reshape(shape)}
        else:
            state[i] = {"step": torch.tensor(float(st["steps"])), "exp_avg": st["a"][off:off + cnt].clone().reshape(shape),
                        "exp_avg_sq": st["b"][off:off + cnt].clone().reshape(shape)}
    if st["kind"] == "sgd":
        group = {"lr": st["lr"] * 100, "momentum": st["momentum"], "dampening": 0, "weight_decay": st["weight_decay"], "nesterov": False,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None, "params": list(range(19))}
    else:
        group = {"lr": st["lr"], "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": st["weight_decay"], "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(19))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(tr, sd) -> None:
    import torch
    from .trainer import param_slots, NUM_PARAMS
    a, b, steps = torch.zeros(NUM_PARAMS), torch.zeros(NUM_PARAMS), 0
    for i, (name, off, cnt) in enumerate(param_slots()):
        e = sd["state"].get(i)
        if e is None:
            continue
        if "momentum_buffer" in e:
            if e["momentum_buffer"] is not None:
                a[off:off + cnt] = e["momentum_buffer"].reshape(-1).float().cpu()
                steps = max(steps, 1)
        else:
            a[off:off + cnt] = e["exp_avg"].reshape(-1).float().cpu()
            b[off:off + cnt] = e["exp_avg_sq"].reshape(-1).float().cpu()
            steps = max(steps, int(float(e["step"])))
    tr.load_optimizer_state({"steps": steps, "a": a, "b": b})


def save_checkpoint(tr, epoch: int, root: str, exp_name: str) -> None:
    import torch
    ckpt = {'epoch': epoch, 'state_dict': {'module.' + k: v for k, v in tr.state_dict().items()}, 'optimizer': optimizer_state_dict(tr)}
    d = os.path.join(root, 'checkpoints', exp_name, 'models')
    os.makedirs(d, exist_ok=True)
    torch.save(ckpt, os.path.join(d, 'epoch_%d.t7' % epoch))
    torch.save(ckpt, os.path.join(d, 'last.t7'))


def initial_state(seed: int) -> Dict[str, np.ndarray]:
    """torch's default initialisation of the reference's module tree under manual_seed(seed) (train.py:268-269, model.py:658-683)"""
    import torch
    from .model import SegModel
    torch.manual_seed(seed)
    net = SegModel(exp_name='init', data_root=os.devnull)
    return {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}


def run_worker(rank: int, world: int, args, make_trainer: Optional[Callable] = None, stage: Optional[Callable] = None,
               init_dist: bool = True) -> Optional[dict]:
    """One rank's training loop.  `make_trainer(state) -> trainer` and `stage(name) -> scene` replace the HIP trainer and the scene
    loader in the CPU (gloo) tests of the driver logic."""
    import torch
    import torch.distributed as dist

    io = IOStream(os.path.join(args.root, 'checkpoints', args.exp_name, 'run.log')) if rank == 0 else None
    if world > 1 and init_dist and not dist.is_initialized():
        dist.init_process_group(backend=args.backend, init_method=f'tcp://127.0.0.1:{args.port}', world_size=world, rank=rank)
    with open(os.path.join(args.root, 'dataset', 'scannet', 'scannetv2_train.txt')) as f:
        scene_list = f.readlines()
    names = [s[:-1] for s in scene_list]
    formats = tuple(x for x in args.out_format.split(',') if x)

    dev = None
    if make_trainer is None:
        from . import cache
        from .scene import DeviceScene
        from .trainer import Trainer
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))       # before the first collective: RCCL binds to the current device
        dev = torch.device('cuda', torch.cuda.current_device())
        if not args.no_cache:
            per_rank = max(1, -(-int(args.workers) // max(world, 1)))
            cache.build_missing(args.root, names[rank::world], args.label_style, workers=per_rank)
            if world > 1:
                dist.barrier()                             # every rank reads packs other ranks built

        def stage(name):   # noqa: F811
            if args.no_cache:
                return DeviceScene.from_reference_tree(name, root=args.root, label_style=args.label_style, device=dev)
            return cache.load_pack(cache.pack_scene(args.root, name, args.label_style), device=dev)

        def make_trainer(state):   # noqa: F811
            first = stage(names[0])
            caps = (max(first.N, 150000), max(first.S, 4096), max(first.E0, 1 << 20), max(first.V, 400000))
            if args.scenes_per_step > 1:
                from .trainer import BatchTrainer
                return BatchTrainer(state, caps, lanes=args.scenes_per_step, device=dev, use_sgd=args.use_sgd, lr=args.lr, momentum=args.momentum,
                                    seed=args.seed + rank)
            return Trainer(state, caps, device=dev, use_sgd=args.use_sgd, lr=args.lr, momentum=args.momentum, seed=args.seed + rank)

    state = initial_state(args.seed)
    start_epoch = 0
    ckpt = None
    if args.resume:
        path = os.path.join(args.root, 'checkpoints', args.exp_name, 'models', 'last.t7')
        if not os.path.exists(path):
            if rank == 0:
                io.cprint('No checkpoint model, please make sure that you use right name in --exp_name')
            raise SystemExit(1)
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        state = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in ckpt['state_dict'].items()}
        start_epoch = int(ckpt['epoch'])
        if rank == 0:
            io.cprint('Load model from ' + path)
    tr = make_trainer(state)
    if ckpt is not None:
        load_optimizer_state_dict(tr, ckpt['optimizer'])
    if rank == 0:
        io.cprint('Network parameters: {}'.format(147880))

    writer = None
    if formats and dev is not None:
        from .model import AsyncLabelWriter
        writer = AsyncLabelWriter(threads=max(2, int(args.workers) // max(world, 1)))
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=2)
    result = None
    for epoch in range(start_epoch, args.epochs):
        tag = 'last' if epoch == args.epochs - 1 else str(epoch + 1)              # train.py:135-138
        mine = epoch_indices(len(names), rank, world, epoch)
        B_step = max(int(getattr(args, 'scenes_per_step', 1)), 1)
        if args.max_steps:
            mine = mine[:args.max_steps * B_step]                                   # --max-steps counts OPTIMIZER steps: B scenes each
        log = EpochLog()
        if getattr(args, 'scenes_per_step', 1) > 1:
            # B scenes per optimizer step (BatchTrainer): the next group is staged while this one trains
            B = args.scenes_per_step
            groups = [mine[k:k + B] for k in range(0, len(mine), B)]
            nxt = [pool.submit(stage, names[si]) for si in groups[0]] if groups else []
            seen = 0
            for gi, grp in enumerate(groups):
                scs = [f.result() for f in nxt]
                nxt = [pool.submit(stage, names[si]) for si in groups[gi + 1]] if gi + 1 < len(groups) else []
                if hasattr(tr, "fits") and any(not tr.fits(sc) for sc in scs):
                    bigger = tuple(max(a, *b) for a, b in zip(tr.caps, zip(*[(sc.N, sc.S, sc.E0, sc.V) for sc in scs])))
                    state_now, opt_now = tr.state_dict(), tr.optimizer_state()
                    gens = tr.generator_states() if hasattr(tr, "generator_states") else None
                    tr.close()
                    from .trainer import BatchTrainer
                    tr = BatchTrainer(state_now, bigger, lanes=B, device=dev, use_sgd=args.use_sgd, lr=args.lr, momentum=args.momentum, seed=args.seed + rank)
                    tr.load_optimizer_state(opt_now)
                    if gens is not None:
                        tr.load_generator_states(gens)                               # the dropout streams go on where they were, not from step 0
                try:
                    _, ress, summed = tr.step(scs)
                except Exception as e:
                    raise RuntimeError('%s: %s' % (' '.join(names[si] for si in grp), e)) from e
                if writer is not None:
                    for si, res in zip(grp, ress):
                        writer.submit(os.path.join(args.root, 'results', args.exp_name, names[si], 'epoch_' + tag), res, formats)
                seen += len(grp)
                if rank == 0:
                    log.add(summed)
                    io.cprint(log.line('Epoch[%d/%d](%04d/%04d)' % (epoch + 1, args.epochs, min(seen * world, len(names)), len(names))))
            mine = []
        nxt = pool.submit(stage, names[mine[0]]) if mine else None
        for i, si in enumerate(mine):
            sc = nxt.result()
            nxt = pool.submit(stage, names[mine[i + 1]]) if i + 1 < len(mine) else None      # the next scene is staged while this one trains
            if hasattr(tr, "fits") and not tr.fits(sc):
                # a scene beyond the trainer's capacities: same parameters / optimizer state / BatchNorm buffers, larger buffers
                bigger = tuple(max(a, b) for a, b in zip(tr.caps, (sc.N, sc.S, sc.E0, sc.V)))
                # the old trainer's device buffers go first: two pipelines of the larger size need not fit together
                state_now, opt_now = tr.state_dict(), tr.optimizer_state()
                gens = tr.generator_states() if hasattr(tr, "generator_states") else None
                tr.close()
                tr = Trainer(state_now, bigger, device=dev, use_sgd=args.use_sgd, lr=args.lr, momentum=args.momentum, seed=args.seed + rank)
                tr.load_optimizer_state(opt_now)
                if gens is not None:
                    tr.load_generator_states(gens)
            try:
                loss, res, summed = tr.step(sc)
            except Exception as e:                                                   # e.g. a scene with one weak instance: BatchNorm1d raises
                raise RuntimeError('%s: %s' % (names[si], e)) from e
            if writer is not None:
                out_root = os.path.join(args.root, 'results', args.exp_name, names[si], 'epoch_' + tag)
                writer.submit(out_root, res, formats)
            if rank == 0:
                log.add(summed)
                io.cprint(log.line('Epoch[%d/%d](%04d/%04d)' % (epoch + 1, args.epochs, (i + 1) * world, len(names))))
        if writer is not None:
            writer.flush()
        if rank == 0:
            io.cprint(log.line('==> Epoch[%d/%d]       ' % (epoch + 1, args.epochs)))
            log.class_report(io)
            save_checkpoint(tr, epoch + 1, args.root, args.exp_name)
            result = dict(epoch=epoch + 1, loss=log.v[0] / max(log.v[165], 1.0), scenes=int(log.v[165]))
    pool.shutdown()
    if writer is not None:
        writer.close()
    if args.param_digests and hasattr(tr, "params"):
        import hashlib
        d = os.path.join(args.root, 'checkpoints', args.exp_name, 'models')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'rank%d.sha256' % rank), 'w') as f:
            f.write(hashlib.sha256(tr.params.detach().cpu().numpy().tobytes()).hexdigest() + '\n')
    if rank == 0:
        io.close()
    if world > 1 and init_dist:
        dist.barrier()
        dist.destroy_process_group()
    if hasattr(tr, "close"):
        tr.close()
    return result


def _spawn_entry(rank, world, args):
    run_worker(rank, world, args)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.resume and args.exp_name is None:
        print("Please choose a specific experiment to resume by using '--exp_name'")      # train.py:249-251
        raise SystemExit(1)
    if args.exp_name is None:
        args.exp_name = time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime())
    import torch
    if args.no_cuda or not torch.cuda.is_available():
        print('seggroup_amd runs on MI355X only: no CPU fallback (use oracle/ for testing)')
        raise SystemExit(1)
    np.seterr(divide='ignore', invalid='ignore')
    for d in ('checkpoints', os.path.join('checkpoints', args.exp_name), os.path.join('checkpoints', args.exp_name, 'models'), 'results',
              os.path.join('results', args.exp_name)):
        os.makedirs(os.path.join(args.root, d), exist_ok=True)                          # train.py:_init_
    if int(os.environ.get('RANK', '0')) == 0:                              # once, like the reference (under torchrun every rank runs main())
        io = IOStream(os.path.join(args.root, 'checkpoints', args.exp_name, 'run.log'))
        io.cprint(str(args))
        io.cprint("Let's use " + str(torch.cuda.device_count()) + " GPUs!")
        io.close()
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
