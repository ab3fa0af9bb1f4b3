"""Drop-in for the reference's `seggroup/util.py`: `cross_entropy_loss` (12-29), `square_loss` (32-38), `IOStream` (41-51).

`cross_entropy_loss` runs on HIP (`sg_cross_entropy_forward` / `_backward`) and carries an autograd node, so a caller's
`loss.backward()` reaches `pred`.  Inside `SegModel.forward` the same loss is fused into the train tail's kernels
(csrc/kernels_train.hip); this entry point exists for code that calls the reference's function by name.
"""
import ctypes as C
import os

import torch

from . import hip


class _SmoothedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gold, smoothing):
        if not pred.is_cuda:
            raise RuntimeError("seggroup_amd.util.cross_entropy_loss needs CUDA/HIP tensors: there is no CPU path")
        lib = hip.lib()
        logits = pred.detach().contiguous().float()
        g = gold.contiguous().view(-1).to(torch.int32)
        K, Cn = int(logits.shape[0]), int(logits.shape[1])
        prob = torch.empty_like(logits)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        s = torch.cuda.current_stream(pred.device).cuda_stream
        hip.check(lib.sg_cross_entropy_forward(logits.data_ptr(), K, Cn, g.data_ptr(), int(bool(smoothing)), prob.data_ptr(), loss.data_ptr(), s))
        ctx.save_for_backward(prob, g)
        ctx.smoothing = int(bool(smoothing))
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        prob, g = ctx.saved_tensors
        lib = hip.lib()
        K, Cn = int(prob.shape[0]), int(prob.shape[1])
        out = torch.empty_like(prob)
        s = torch.cuda.current_stream(prob.device).cuda_stream
        hip.check(lib.sg_cross_entropy_backward(prob.data_ptr(), K, Cn, g.data_ptr(), ctx.smoothing, C.c_float(float(gout)), out.data_ptr(), s))
        return out, None, None


def cross_entropy_loss(pred, gold, smoothing=True):
    """Calculate cross entropy loss (sum over the rows), apply label smoothing (eps = 0.2) if needed.  util.py:12-29"""
    return _SmoothedCE.apply(pred, gold, smoothing)


def square_loss(euclidean_distance):
    """Square loss function (util.py:32-38; nothing in the reference calls it): one fused torch reduction."""
    return torch.sum(torch.pow(euclidean_distance, 2))


class IOStream:
    """print + append + flush (reference seggroup/util.py:41-51)."""

    def __init__(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        self.f = open(path, 'a')

    def cprint(self, text):
        print(text)
        self.f.write(text + '\n')
        self.f.flush()

    def close(self):
        self.f.close()
