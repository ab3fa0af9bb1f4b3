"""Training driver on CPU (multi-GPU part): world_size-2 gloo processes run seggroup_amd.train.run_worker with a stub
trainer -- the sampler order, the ONE all-reduce per step (gradient averaged, log terms summed), the log lines and the checkpoint
layout are host logic; the HIP step itself is covered by tests/test_gpu_train.py."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

os.environ.setdefault("SEGGROUP_HOST_ONLY", "1")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubScene:
    def __init__(self, name):
        self.name = name
        self.index = int(name[5:9])


class _StubTrainer:
    """Same surface as seggroup_amd.trainer.Trainer, CPU tensors: the 'gradient' of scene i is the constant i + 1, the optimizer is
    plain SGD with lr 1 -- after a step every rank must hold params - mean over the ranks' scenes of (i + 1)."""

    def __init__(self, state):
        import torch
        from seggroup_amd import trainer as T
        self.T = T
        self.params = torch.from_numpy(T.flatten_state(state)).clone()
        self.grads_full = torch.zeros(T.NUM_PARAMS + T.NUM_EXTRAS)
        self.buffers = {}
        self.steps = 0
        self.seen = []
        self.opt_a, self.opt_b = torch.zeros(T.NUM_PARAMS), torch.zeros(T.NUM_PARAMS)

    def step(self, sc, keep="random"):
        from types import SimpleNamespace
        T = self.T
        self.grads_full[:T.NUM_PARAMS] = float(sc.index + 1)
        rng = np.random.default_rng(sc.index)
        res = SimpleNamespace(iou_sem=rng.integers(1, 9, (1, 2, 40)).astype(np.float32), iou_ins=rng.integers(1, 9, (1, 2, 40)).astype(np.float32),
                              acc=rng.uniform(size=4).astype(np.float32))
        loss = np.array([[2.0 * (sc.index + 1), 2.0]], np.float32)
        extras = np.concatenate([[loss[0, 0] / loss[0, 1]], res.iou_sem.reshape(-1), res.iou_ins.reshape(-1), res.acc, [1.0]]).astype(np.float32)
        summed = T.allreduce_step(self.grads_full, extras)
        self.opt_a = 0.9 * self.opt_a + self.grads_full[:T.NUM_PARAMS] if self.steps else self.grads_full[:T.NUM_PARAMS].clone()
        self.params -= self.grads_full[:T.NUM_PARAMS]
        self.steps += 1
        self.seen.append(sc.index)
        return loss, res, summed

    def state_dict(self):
        return self.T.Trainer.state_dict(self)

    def optimizer_state(self):
        return {"kind": "sgd", "steps": self.steps, "a": self.opt_a, "b": self.opt_b, "lr": 0.001, "momentum": 0.9, "weight_decay": 1e-4}

    def load_optimizer_state(self, st):
        self.steps, self.opt_a, self.opt_b = int(st["steps"]), st["a"].clone(), st["b"].clone()


class _StubBatchTrainer(_StubTrainer):
    """seggroup_amd.trainer.BatchTrainer's surface: a step takes a LIST of scenes, its gradient is the mean of theirs, the log terms their sum"""

    def fits(self, sc):
        return True

    def step(self, scs, keep="random"):
        from types import SimpleNamespace
        T = self.T
        self.grads_full[:T.NUM_PARAMS] = float(np.mean([sc.index + 1 for sc in scs]))
        extras = np.zeros(165, np.float32)
        ress, losses = [], []
        for sc in scs:
            rng = np.random.default_rng(sc.index)
            res = SimpleNamespace(iou_sem=rng.integers(1, 9, (1, 2, 40)).astype(np.float32), iou_ins=rng.integers(1, 9, (1, 2, 40)).astype(np.float32),
                                  acc=rng.uniform(size=4).astype(np.float32))
            loss = np.array([[2.0 * (sc.index + 1), 2.0]], np.float32)
            extras += np.concatenate([[loss[0, 0] / loss[0, 1]], res.iou_sem.reshape(-1), res.iou_ins.reshape(-1), res.acc]).astype(np.float32)
            ress.append(res); losses.append(loss)
            self.seen.append(sc.index)
        summed = T.allreduce_step(self.grads_full, np.concatenate([extras, [float(len(scs))]]).astype(np.float32))
        self.params -= self.grads_full[:T.NUM_PARAMS]
        self.steps += 1
        return losses, ress, summed


def _worker(rank, world, root, port, q, resume, per_step=1):
    sys.path.insert(0, ROOT)
    import torch
    from seggroup_amd import train, trainer as T
    argv = ["-n", "exp", "--root", root, "--backend", "gloo", "--port", str(port), "--epochs", "2", "--out-format", ""] + (["-r"] if resume else [])
    if per_step > 1:
        argv += ["--scenes-per-step", str(per_step)]
    args = train.build_parser().parse_args(argv)
    made = []

    def make(state):
        tr = (_StubBatchTrainer if per_step > 1 else _StubTrainer)(state)
        for name, _, n in T.BN_LAYERS:
            tr.buffers[name + ".running_mean"] = torch.zeros(n)
            tr.buffers[name + ".running_var"] = torch.ones(n)
            tr.buffers[name + ".num_batches_tracked"] = torch.tensor(0)
        made.append(tr)
        return tr
    r = train.run_worker(rank, world, args, make_trainer=make, stage=_StubScene)
    q.put((rank, made[0].seen, float(made[0].params[0]), float(made[0].params[-1]), r))


def _run(root, resume=False, world=2, per_step=1):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, root, port, q, resume, per_step)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_sampler_order_is_distributed_samplers():
    import torch
    from torch.utils.data.distributed import DistributedSampler
    from seggroup_amd import train
    for n, world in ((11, 2), (1201, 8)):
        for epoch in (0, 3):
            for rank in range(world):
                s = DistributedSampler(range(n), num_replicas=world, rank=rank)          # train.py:101 (shuffle=True, seed 0)
                s.set_epoch(epoch)
                assert list(s) == train.epoch_indices(n, rank, world, epoch)


def test_two_rank_training_loop_and_checkpoint_layout(tmp_path):
    import torch
    from seggroup_amd import train, trainer as T
    from seggroup_amd.model import SegModel
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "dataset", "scannet"))
    n = 6
    with open(os.path.join(root, "dataset", "scannet", "scannetv2_train.txt"), "w") as f:
        f.write("".join(f"scene{i:04d}_00\n" for i in range(n)))
    got = _run(root)
    # every rank saw its DistributedSampler share, epoch after epoch
    for rank, seen, p0, p1, _ in got:
        assert seen == train.epoch_indices(n, rank, 2, 0) + train.epoch_indices(n, rank, 2, 1)
    # identical parameters on both ranks = initial - sum over steps of the MEAN of the two ranks' gradients
    init = T.flatten_state(train.initial_state(1))
    mean_steps = sum((a + 1 + b + 1) / 2.0 for e in (0, 1) for a, b in zip(train.epoch_indices(n, 0, 2, e), train.epoch_indices(n, 1, 2, e)))
    for _, _, p0, p1, _ in got:
        assert abs(p0 - (init[0] - mean_steps)) < 1e-3 and abs(p1 - (init[-1] - mean_steps)) < 1e-3
    res = got[0][4]
    assert res["epoch"] == 2 and res["scenes"] == n
    # the log carries the reference's lines; the loss column is the mean over all scenes of the epoch of loss_sum / K = index + 1
    log = open(os.path.join(root, "checkpoints", "exp", "run.log")).read()
    assert "Epoch[1/2](0002/0006)    Loss:" in log and "==> Epoch[2/2]" in log and "Semantic mIoU (20 classes)" in log
    assert "==> Epoch[2/2]           Loss: %.6f" % np.mean([i + 1 for i in range(n)]) in log
    # checkpoint: DDP-style keys that load STRICTLY into the module tree (with the Sequential aliases) after stripping 'module.',
    # and an optimizer state torch.optim.SGD accepts over named_parameters() order
    ck = torch.load(os.path.join(root, "checkpoints", "exp", "models", "last.t7"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 2 and os.path.exists(os.path.join(root, "checkpoints", "exp", "models", "epoch_1.t7"))
    net = SegModel(exp_name="x", data_root=os.devnull)
    net.load_state_dict({k[len("module."):]: v for k, v in ck["state_dict"].items()}, strict=True)
    assert [k for k, _ in net.named_parameters()] == [nm for nm, _, _ in T.param_slots()]
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    opt.load_state_dict(ck["optimizer"])
    assert opt.state_dict()["state"][18]["momentum_buffer"].shape == (40,)
    # --resume continues from the stored epoch: nothing left to do at epochs == 2, parameters come from the checkpoint
    again = _run(root, resume=True)
    for _, seen, p0, _, r in again:
        assert seen == [] and abs(p0 - got[0][2]) < 1e-6


def test_allreduce_step_single_rank_is_identity():
    import torch
    from seggroup_amd import trainer as T
    g = torch.arange(T.NUM_PARAMS + T.NUM_EXTRAS, dtype=torch.float32)
    out = T.allreduce_step(g, np.array([1.0, 2.0], np.float32))
    assert out.tolist() == [1.0, 2.0] and float(g[5]) == 5.0
    with pytest.raises(ValueError):
        T.allreduce_step(g, np.zeros(T.NUM_EXTRAS + 1, np.float32))


def test_two_ranks_with_two_scenes_per_step(tmp_path):
    """--scenes-per-step 2 (BatchTrainer's driver loop) on two gloo ranks: every rank walks its DistributedSampler share in groups of two
    (the last group of an epoch has one scene), one all-reduce per group, the log counts scenes, both ranks end with the same parameters."""
    from seggroup_amd import train, trainer as T
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "dataset", "scannet"))
    n = 6
    with open(os.path.join(root, "dataset", "scannet", "scannetv2_train.txt"), "w") as f:
        f.write("".join(f"scene{i:04d}_00\n" for i in range(n)))
    got = _run(root, per_step=2)
    for rank, seen, _, _, _ in got:
        assert seen == train.epoch_indices(n, rank, 2, 0) + train.epoch_indices(n, rank, 2, 1)
    init = T.flatten_state(train.initial_state(1))
    moved = 0.0
    for e in (0, 1):
        shares = [train.epoch_indices(n, r, 2, e) for r in (0, 1)]
        for k in range(0, 3, 2):                                   # groups [0:2], [2:3] of each rank's three scenes
            moved += np.mean([np.mean([i + 1 for i in sh[k:k + 2]]) for sh in shares])      # DDP mean over ranks of the lanes' mean
    for _, _, p0, p1, _ in got:
        assert abs(p0 - (init[0] - moved)) < 1e-3 and abs(p1 - (init[-1] - moved)) < 1e-3
    res = got[0][4]
    assert res["epoch"] == 2 and res["scenes"] == n
    log = open(os.path.join(root, "checkpoints", "exp", "run.log")).read()
    assert "Epoch[1/2](0004/0006)    Loss:" in log and "Epoch[1/2](0006/0006)    Loss:" in log and "==> Epoch[2/2]" in log
