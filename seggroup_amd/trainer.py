"""One training step per scene on HIP: host side of `sg_trainer` (csrc/trainer.cpp).

Mirrors the loop body of the reference's train.py:160-170 --

    loss_raw, IoU_sem, IoU_ins, acc = model(data, weak_label, info)
    loss = loss_raw[:, 0].sum() / loss_raw[:, 1].sum();  optimizer.zero_grad();  loss.backward();  optimizer.step()

-- with the parameters, the gradient and the optimizer state as FLAT device vectors in the reference's
`named_parameters()` order (`param_slots()`), so that the DistributedDataParallel gradient averaging of train.py:88 is ONE
all-reduce of 0.59 MB over RCCL and the optimizer is one kernel.  `state_dict()` / `load_state_dict()` speak the reference's
checkpoint keys, including the running BatchNorm statistics the reference updates in training mode.
"""
import contextlib
import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import hip
from .model import SceneResult
from .scene import DeviceScene

NUM_PARAMS = 147880
NUM_EXTRAS = 168          # floats riding behind the gradient in the same all-reduce: loss, 2 x 2 x 40 IoU sums, 4 accuracies, a count
BN_LAYERS = (("mlp_1.bn1", 0, 64), ("mlp_2.bn1", 128, 64), ("mlp_3.bn1", 256, 64), ("mlp_3.bn2", 384, 64), ("classifier.bn1", 512, 128))


def param_slots():
    """[(name, offset, count)] of the flat parameter vector = the reference model's named_parameters() order (host-only call)."""
    lib = hip.lib()
    out = []
    for i in range(19):
        name, off, cnt = C.c_char_p(), C.c_int(), C.c_int()
        hip.check(lib.sg_param_slot(i, C.byref(name), C.byref(off), C.byref(cnt)))
        out.append((name.value.decode(), off.value, cnt.value))
    return out


PARAM_SHAPES = {"mlp_1.conv1.0.weight": (64, 6, 1, 1), "mlp_2.conv1.0.weight": (64, 18, 1, 1), "mlp_3.conv1.0.weight": (64, 18, 1, 1),
                "mlp_3.conv2.0.weight": (64, 64, 1, 1), "gcn_2.fc.weight": (192, 192), "gcn_3.fc.weight": (256, 256),
                "classifier.linear1.weight": (128, 256), "classifier.linear2.weight": (40, 128)}


def flatten_state(state: Dict[str, "np.ndarray | torch.Tensor"]) -> np.ndarray:
    """state_dict (reference keys) -> flat float32 vector; a missing key raises like load_state_dict(strict=True)"""
    flat = np.zeros(NUM_PARAMS, np.float32)
    for name, off, cnt in param_slots():
        if name not in state:
            raise KeyError(f"missing parameter {name!r}")
        v = state[name]
        v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if v.size != cnt:
            raise ValueError(f"{name}: {v.size} values, expected {cnt}")
        flat[off:off + cnt] = v.reshape(-1)
    return flat


def allreduce_step(grads_full: torch.Tensor, extras: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
    """grads_full = [gradient (NUM_PARAMS) | extras (NUM_EXTRAS)]: all-reduce (sum) once, then the gradient part is divided by the
    world size (DDP averages, train.py:88) and the summed extras are returned.  A no-op on one rank."""
    import torch.distributed as dist
    n = 0 if extras is None else int(np.asarray(extras).size)
    if n > NUM_EXTRAS:
        raise ValueError(f"{n} extras, room for {NUM_EXTRAS}")
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return None if extras is None else np.asarray(extras, np.float32).copy()
    tail = grads_full[NUM_PARAMS:]
    tail.zero_()
    if n:
        tail[:n] = torch.as_tensor(np.asarray(extras, np.float32).reshape(-1)).to(tail.device)
    dist.all_reduce(grads_full)
    grads_full[:NUM_PARAMS].div_(world)
    return tail[:n].cpu().numpy() if n else None


class Trainer:
    """`step(scene)` = forward + loss + backward + gradient averaging over the ranks + optimizer step, all on the device.

    use_sgd / lr / momentum follow train.py:95-99: SGD(lr * 100, momentum, weight_decay 1e-4) or Adam(lr, weight_decay 1e-4)."""

    def __init__(self, state: Dict[str, np.ndarray], caps, device=None, use_sgd: bool = True, lr: float = 0.001, momentum: float = 0.9,
                 weight_decay: float = 1e-4, seed: int = 1, params: Optional[torch.Tensor] = None, own_stream: bool = False):
        """`params`: a flat device vector to train IN PLACE of a private copy of `state` (the lanes of a BatchTrainer share one);
        `own_stream`: forward / loss / backward on a stream of this trainer's own instead of the default stream (lanes run side by side)."""
        hip.require_device()
        self.lib = hip.lib()
        self.device = torch.device(device if device is not None else "cuda")
        self.caps = tuple(int(c) for c in caps)
        self.params = params if params is not None else torch.from_numpy(flatten_state(state)).to(self.device)
        self.stream = torch.cuda.Stream(device=self.device) if own_stream else None
        # gradient vector + the per-step log quantities of train.py:170-173 behind it: ONE collective per step
        self.grads_full = torch.zeros(NUM_PARAMS + NUM_EXTRAS, dtype=torch.float32, device=self.device)
        self.grads = self.grads_full[:NUM_PARAMS]
        self.use_sgd, self.lr, self.momentum, self.weight_decay = bool(use_sgd), float(lr), float(momentum), float(weight_decay)
        self.opt_a = torch.zeros_like(self.params)          # SGD momentum buffer / Adam exp_avg
        self.opt_b = torch.zeros_like(self.params)          # Adam exp_avg_sq
        self.steps = 0
        self.buffers: Dict[str, torch.Tensor] = {}
        host = lambda v: None if v is None else (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
        state = {k: host(v) for k, v in state.items()}
        for name, _, n in BN_LAYERS:
            rm, rv = state.get(name + ".running_mean"), state.get(name + ".running_var")
            self.buffers[name + ".running_mean"] = torch.as_tensor(np.asarray(rm, np.float32) if rm is not None else np.zeros(n, np.float32)).clone()
            self.buffers[name + ".running_var"] = torch.as_tensor(np.asarray(rv, np.float32) if rv is not None else np.ones(n, np.float32)).clone()
            nb = state.get(name + ".num_batches_tracked")
            self.buffers[name + ".num_batches_tracked"] = torch.tensor(int(np.asarray(nb)) if nb is not None else 0, dtype=torch.int64)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        self.labels = torch.empty((hip.NUM_LABEL_VECTORS, self.caps[3]), dtype=torch.int32, pin_memory=True)
        with torch.cuda.device(self.device):
            self.handle = self.lib.sg_trainer_create(*self.caps, self.params.data_ptr(), self.grads.data_ptr(),
                                                     C.c_void_p(self.stream.cuda_stream) if self.stream is not None else None)
        if not self.handle:
            raise hip.SgError(hip.SG_EHIP, self.lib.sg_last_error().decode())
        self.K = self.C5 = 0

    # ---- the three phases ------------------------------------------------------------------------------------------
    def fits(self, sc: DeviceScene) -> bool:
        return sc.N <= self.caps[0] and sc.S <= self.caps[1] and sc.E0 <= self.caps[2] and sc.V <= self.caps[3]

    def forward(self, sc: DeviceScene) -> SceneResult:
        """SegModel.forward up to the classifier with the current parameters (pseudo labels + metrics as in ins_infer)."""
        res = hip.Result()
        res.h_labels = self.labels.data_ptr()
        c5, k = C.c_int(), C.c_int()
        with torch.cuda.device(self.device):
            hip.check(self.lib.sg_trainer_forward(self.handle, C.byref(sc.c_struct), C.byref(res), C.byref(c5), C.byref(k)))
        self.C5, self.K = c5.value, k.value
        lab = self.labels.numpy().reshape(-1)[:hip.NUM_LABEL_VECTORS * sc.V].reshape(hip.NUM_LABEL_VECTORS, sc.V).copy()
        return SceneResult(lab, 14, res)

    def dropout_mask(self, keep="random") -> Optional[torch.Tensor]:
        """Dropout(p=0.5) keep mask [K,128] scaled by 2: "random" from this trainer's generator, "pinned" = the counter-based mask
        of the golden capture, None = no dropout, or a ready mask."""
        if keep is None:
            return None
        if isinstance(keep, str) and keep == "random":
            with torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext():
                return (torch.rand((self.K, 128), device=self.device, generator=self.gen) >= 0.5).float() * 2.0
        if isinstance(keep, str) and keep == "pinned":
            from .synthetic import uniform01
            keep = np.where(uniform01(97, self.K, self.K * 128).reshape(self.K, 128) < 0.5, 2.0, 0.0).astype(np.float32)
        with torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext():
            return torch.as_tensor(np.asarray(keep, np.float32) if not isinstance(keep, torch.Tensor) else keep).to(self.device).float().contiguous()

    def loss(self, mask: Optional[torch.Tensor], want_logits: bool = False):
        """-> loss [1,2] = [[loss_sum, K]] (model.py:928-930)"""
        h = (C.c_float * 2)()
        self.logits = torch.empty((self.K, 40), dtype=torch.float32, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            hip.check(self.lib.sg_trainer_loss(self.handle, hip.ptr(mask), h, hip.ptr(self.logits)))
        return np.array([[h[0], h[1]]], dtype=np.float32)

    def backward(self, mask: Optional[torch.Tensor], scale: float = 0.0) -> torch.Tensor:
        """gradient of scale * loss_sum (default 1 / K) into `self.grads` (flat, overwritten)"""
        with torch.cuda.device(self.device):
            hip.check(self.lib.sg_trainer_backward(self.handle, hip.ptr(mask), C.c_float(scale)))
        return self.grads

    def average_gradients(self, extras: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """DistributedDataParallel's gradient averaging (train.py:88) as one all-reduce of the flat vector over RCCL; `extras`
        (<= NUM_EXTRAS floats: this rank's loss / IoU / accuracy terms, train.py:170-173) ride in the same collective and come
        back SUMMED over the ranks."""
        return allreduce_step(self.grads_full, extras)

    def optimizer_step(self) -> None:
        self.steps += 1
        n = NUM_PARAMS
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream().cuda_stream
            if self.use_sgd:
                hip.check(self.lib.sg_optimizer_sgd(self.params.data_ptr(), self.grads.data_ptr(), self.opt_a.data_ptr(), n, C.c_float(self.lr * 100),
                                                    C.c_float(self.momentum), C.c_float(self.weight_decay), int(self.steps == 1), s))
            else:
                hip.check(self.lib.sg_optimizer_adam(self.params.data_ptr(), self.grads.data_ptr(), self.opt_a.data_ptr(), self.opt_b.data_ptr(), n,
                                                     C.c_float(self.lr), C.c_float(self.weight_decay), self.steps, s))

    def update_running_stats(self) -> None:
        """BatchNorm's running_mean / running_var (momentum 0.1, unbiased variance) + num_batches_tracked, as the reference's
        modules do in training mode"""
        rows = (C.c_double * 5)()
        buf = (C.c_float * 768)()
        hip.check(self.lib.sg_trainer_bn_stats(self.handle, buf, rows))
        stats = np.frombuffer(buf, dtype=np.float32)
        self.last_bn_stats = stats.copy()
        for i, (name, off, n) in enumerate(BN_LAYERS):
            mean, var = stats[off:off + n], stats[off + n:off + 2 * n]
            r = float(rows[i])
            self.buffers[name + ".running_mean"].mul_(0.9).add_(torch.from_numpy(mean.copy()) * 0.1)
            self.buffers[name + ".running_var"].mul_(0.9).add_(torch.from_numpy(var.copy()) * (0.1 * r / max(r - 1.0, 1.0)))
            self.buffers[name + ".num_batches_tracked"] += 1

    def step(self, sc: DeviceScene, keep="random"):
        """One iteration of train.py:160-173 on this rank's scene -> (loss [1,2], SceneResult, summed log terms).
        Log terms (summed over the ranks): [loss_sum / K, IoU_sem (2 x 40), IoU_ins (2 x 40), acc (4), 1]."""
        res = self.forward(sc)
        mask = self.dropout_mask(keep)
        loss = self.loss(mask)
        self.backward(mask)
        extras = np.concatenate([[loss[0, 0] / loss[0, 1]], res.iou_sem.reshape(-1), res.iou_ins.reshape(-1), res.acc.reshape(-1), [1.0]]).astype(np.float32)
        summed = self.average_gradients(extras)
        self.optimizer_step()
        self.update_running_stats()
        return loss, res, summed

    # ---- checkpoint contract -----------------------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        flat = self.params.detach().cpu()
        out = {}
        for name, off, cnt in param_slots():
            out[name] = flat[off:off + cnt].clone().reshape(PARAM_SHAPES.get(name, (cnt,)))
        out.update({k: v.clone() for k, v in self.buffers.items()})
        # the reference registers every BatchNorm2d twice -- as `bnN` and as element 1 of the `convN` Sequential (model.py:42-44,
        # 88-90,121-126) -- so its state_dict carries both names and load_state_dict(strict=True) wants both
        for blk, n in (("mlp_1", 1), ("mlp_2", 1), ("mlp_3", 1), ("mlp_3", 2)):
            for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
                out[f"{blk}.conv{n}.1.{leaf}"] = out[f"{blk}.bn{n}.{leaf}"].clone()
        return out

    def load_state_dict(self, state) -> None:
        self.params.copy_(torch.from_numpy(flatten_state(state)))
        for k in self.buffers:
            if k in state:
                self.buffers[k] = torch.as_tensor(state[k]).detach().cpu().clone().to(self.buffers[k].dtype)

    def optimizer_state(self) -> Dict[str, object]:
        return {"kind": "sgd" if self.use_sgd else "adam", "steps": self.steps, "a": self.opt_a.detach().cpu(), "b": self.opt_b.detach().cpu(),
                "lr": self.lr, "momentum": self.momentum, "weight_decay": self.weight_decay}

    def load_optimizer_state(self, st) -> None:
        self.steps = int(st["steps"])
        self.opt_a.copy_(st["a"].to(self.opt_a.device)); self.opt_b.copy_(st["b"].to(self.opt_b.device))

    def generator_states(self) -> List[torch.Tensor]:
        """the dropout generator's state (one per lane for a BatchTrainer): carried into the trainer that replaces this one on growth"""
        return [self.gen.get_state()]

    def load_generator_states(self, states) -> None:
        self.gen.set_state(states[0])

    def device_bytes(self) -> int:
        return int(self.lib.sg_trainer_device_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sg_trainer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchTrainer(Trainer):
    """B scenes per optimizer step on ONE GPU: B lanes (an `sg_trainer` each, on its own stream, driven by its own host thread) run forward +
    loss + backward side by side on the SAME parameter vector, their gradients are averaged and one optimizer step follows -- what the reference
    does with B ranks (DistributedDataParallel averages the ranks' gradients, train.py:88; every rank sees its own scene and its own BatchNorm
    batch statistics), without B processes.  One scene's step is a chain of latency-bound launches and host round trips (6.6 ms for 5 ms of
    kernels); B of them overlap.  The running BatchNorm statistics follow lane 0, as DDP's buffer broadcast from rank 0 does.

    `step(scenes)` takes up to B scenes; the ranks' all-reduce (one per step, as before) averages the lane-averaged gradients."""

    def __init__(self, state: Dict[str, np.ndarray], caps, lanes: int = 4, device=None, use_sgd: bool = True, lr: float = 0.001,
                 momentum: float = 0.9, weight_decay: float = 1e-4, seed: int = 1):
        super().__init__(state, caps, device=device, use_sgd=use_sgd, lr=lr, momentum=momentum, weight_decay=weight_decay, seed=seed, own_stream=True)
        from concurrent.futures import ThreadPoolExecutor
        self.lanes: List[Trainer] = [self] + [Trainer(state, caps, device=self.device, seed=seed + 7919 * k, params=self.params, own_stream=True)
                                              for k in range(1, max(1, int(lanes)))]
        self.pool = ThreadPoolExecutor(max_workers=len(self.lanes))

    def generator_states(self) -> List[torch.Tensor]:
        return [lane.gen.get_state() for lane in self.lanes]

    def load_generator_states(self, states) -> None:
        for lane, st in zip(self.lanes, states):
            lane.gen.set_state(st)

    @staticmethod
    def _lane_pass(lane: Trainer, sc: DeviceScene, keep):
        res = lane.forward(sc)
        mask = lane.dropout_mask(keep)
        loss = lane.loss(mask)
        lane.backward(mask)
        lane.stream.synchronize()
        return loss, res

    def forward_backward(self, scenes, keep="random"):
        """The lanes' passes side by side; afterwards `self.grads` holds the MEAN of the scenes' gradients.  -> [(loss [1,2], SceneResult)]"""
        if isinstance(scenes, DeviceScene):
            scenes = [scenes]
        if not 1 <= len(scenes) <= len(self.lanes):
            raise ValueError(f"{len(scenes)} scenes for {len(self.lanes)} lanes")
        torch.cuda.current_stream(self.device).synchronize()          # the optimizer step of the last call wrote the parameters on the default stream
        done = list(self.pool.map(lambda a: self._lane_pass(*a), [(self.lanes[i], sc, keep) for i, sc in enumerate(scenes)]))
        n = len(scenes)
        if n > 1:                                                     # lane 0's vector doubles as the step's gradient: mean over the lanes, in lane order
            for lane in self.lanes[1:n]:
                self.grads.add_(lane.grads)
            self.grads.div_(n)
        return done

    def step(self, scenes, keep="random"):
        """-> ([loss [1,2] per scene], [SceneResult per scene], log terms summed over scenes and ranks)"""
        done = self.forward_backward(scenes, keep)
        n = len(done)
        extras = np.zeros(NUM_EXTRAS - 3, np.float32)                 # [loss / K, IoU_sem (80), IoU_ins (80), acc (4)] summed over the scenes
        for loss, res in done:
            extras += np.concatenate([[loss[0, 0] / loss[0, 1]], res.iou_sem.reshape(-1), res.iou_ins.reshape(-1), res.acc.reshape(-1)]).astype(np.float32)
        summed = self.average_gradients(np.concatenate([extras, [float(n)]]).astype(np.float32))
        self.optimizer_step()
        self.update_running_stats()
        return [d[0] for d in done], [d[1] for d in done], summed

    def close(self):
        for lane in getattr(self, "lanes", [])[1:]:
            lane.close()
        if getattr(self, "pool", None) is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        super().close()
