"""Network parameters of the SegGroup grouping network (147,880 parameters).

Key layout follows the reference checkpoint contract (reference
seggroup/model.py:65-166,676-681 and seggroup/infer.py:112-121): a `state_dict` whose keys may
carry a `module.` prefix (DDP) and in which every BatchNorm appears under two aliases
(`mlp_k.bn1.*` and `mlp_k.conv1.1.*`).  Only the tensors the inference path reads are kept:

  mlp_1.conv1.0.weight [64,6]   mlp_1.bn1.{weight,bias} [64]
  mlp_2.conv1.0.weight [64,18]  mlp_2.bn1.{weight,bias} [64]
  gcn_2.fc.weight      [192,192]
  mlp_3.conv1.0.weight [64,18]  mlp_3.bn1.{weight,bias} [64]
  mlp_3.conv2.0.weight [64,64]  mlp_3.bn2.{weight,bias} [64]
  gcn_3.fc.weight      [256,256]

BatchNorm running statistics are never read: infer.py leaves the model in train() mode
(infer.py:131-136), so every BN normalises with per-scene batch statistics.
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from .synthetic import uniform01

# name -> shape (as the inference path consumes them; conv kernels squeezed to 2-D)
PARAM_SHAPES = {
    "mlp_1.conv1.0.weight": (64, 6),
    "mlp_1.bn1.weight": (64,), "mlp_1.bn1.bias": (64,),
    "mlp_2.conv1.0.weight": (64, 18),
    "mlp_2.bn1.weight": (64,), "mlp_2.bn1.bias": (64,),
    "gcn_2.fc.weight": (192, 192),
    "mlp_3.conv1.0.weight": (64, 18),
    "mlp_3.bn1.weight": (64,), "mlp_3.bn1.bias": (64,),
    "mlp_3.conv2.0.weight": (64, 64),
    "mlp_3.bn2.weight": (64,), "mlp_3.bn2.bias": (64,),
    "gcn_3.fc.weight": (256, 256),
}

_BN_ALIASES = {
    "mlp_1.conv1.1": "mlp_1.bn1",
    "mlp_2.conv1.1": "mlp_2.bn1",
    "mlp_3.conv1.1": "mlp_3.bn1",
    "mlp_3.conv2.1": "mlp_3.bn2",
}


def make_weights(seed: int = 1, bn1_gamma: float = 2.0, affine_jitter: float = 0.0) -> Dict[str, np.ndarray]:
    """Deterministic parameters with torch's default-init distributions.

    conv / linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (kaiming_uniform with a=sqrt(5));
    BN: gamma = 1, beta = 0, except mlp_1.bn1 gamma = `bn1_gamma` (the "gamma x2" set of
    BASELINE.md section 2 that makes all four grouping layers merge).  `affine_jitter` adds
    U(-j, j) to every gamma and beta so tests exercise the affine terms.
    """
    out: Dict[str, np.ndarray] = {}
    for i, (name, shape) in enumerate(PARAM_SHAPES.items()):
        n = int(np.prod(shape))
        u = uniform01(seed, 1000 + i, n).astype(np.float64)
        if name.endswith("bn1.weight") or name.endswith("bn2.weight"):
            base = bn1_gamma if name == "mlp_1.bn1.weight" else 1.0
            w = base + affine_jitter * (2.0 * u - 1.0)
        elif name.endswith(".bias"):
            w = affine_jitter * (2.0 * u - 1.0)
        else:
            bound = 1.0 / np.sqrt(shape[1])
            w = (2.0 * u - 1.0) * bound
        out[name] = w.astype(np.float32).reshape(shape)
    return out


def from_state_dict(sd: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """Extract inference parameters from a reference-style state_dict (torch tensors or arrays)."""
    if "state_dict" in sd and not any(k.endswith("weight") for k in sd):
        sd = sd["state_dict"]  # checkpoint dict {'epoch','state_dict','optimizer'} (train.py:216-220)
    flat = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        for alias, canon in _BN_ALIASES.items():
            if k.startswith(alias + "."):
                k = canon + k[len(alias):]
        flat[k] = v
    out = {}
    for name, shape in PARAM_SHAPES.items():
        if name not in flat:
            raise KeyError(f"checkpoint is missing '{name}'")
        v = flat[name]
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        out[name] = np.ascontiguousarray(a, dtype=np.float32).reshape(shape)
    return out


def to_state_dict(w: Mapping[str, np.ndarray], prefix: str = "module."):
    """Reference-shaped state_dict (torch tensors, 4-D conv kernels, both BN aliases)."""
    import torch

    sd = {}
    for name, a in w.items():
        t = torch.from_numpy(np.array(a, dtype=np.float32))
        if ".conv" in name and name.endswith(".0.weight"):
            t = t.reshape(t.shape[0], t.shape[1], 1, 1)
        sd[prefix + name] = t
    for alias, canon in _BN_ALIASES.items():
        for leaf in ("weight", "bias"):
            sd[prefix + alias + "." + leaf] = sd[prefix + canon + "." + leaf]
    return sd


def to_full_state_dict(w: Mapping[str, np.ndarray], prefix: str = "module."):
    """A COMPLETE reference checkpoint `state_dict` (all 54 entries of SURVEY.md 8b: BN running statistics and counters,
    the train-only classifier at its default init) carrying the inference parameters `w` -- what `train.py:216-220` saves
    and `infer.py:121` loads with strict key matching."""
    from .model import SegModel

    net = SegModel(exp_name="_ckpt", ins_infer=True, data_root="/nonexistent")
    net.load_weights(w)
    return {prefix + k: v.detach().clone() for k, v in net.state_dict().items()}


def save_npz(path: str, w: Mapping[str, np.ndarray]) -> None:
    np.savez_compressed(path, **{k.replace(".", "__"): v for k, v in w.items()})


def load_npz(path: str) -> Dict[str, np.ndarray]:
    z = np.load(path)
    return {k.replace("__", "."): z[k] for k in z.files}
