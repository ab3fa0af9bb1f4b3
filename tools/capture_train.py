#!/usr/bin/env python3
"""Golden vectors of the TRAIN-mode tail from the real reference (build container only; SURVEY.md 8f-4, first slice).

Runs the unmodified reference `SegModel.forward` with neither infer flag set (model.py:900-932) on the small fixtures, with
the classifier's Dropout(p=0.5) replaced -- on the module INSTANCE, no reference file is edited -- by the pinned mask
`oracle.cpu_ref.dropout_keep(K)` (in training mode the reference's own output depends on torch's RNG stream, so parity is
stated with the same mask on both sides), and stores what the tail consumes and produces:

    classifier weights (torch default init under manual_seed(1)), Feat_6, logits, loss [1,2], the pinned mask

into tests/golden/train_tail.npz, and -- the rest of the training step (model.py:900-932 + train.py:164-168) -- the
gradients `loss.backward()` leaves on EVERY parameter of the reference model for loss = loss_sum / loss_num, together with the
running BatchNorm statistics after that one forward, into tests/golden/train_grads.npz (`<fixture>.grad.<parameter>`,
`<fixture>.buf.<buffer>`).  usage: python tools/capture_train.py
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import capture_reference as cap  # noqa: E402


def main():
    tp = cap._install_shims()
    np.seterr(divide="ignore", invalid="ignore")
    import json
    import torch
    import model as model_mod
    model_mod.torch = tp
    from oracle import cpu_ref
    from seggroup_amd import synthetic, weights as W

    wts = W.load_npz(os.path.join(REPO, "tests", "golden", "weights_g2.npz"))
    index = json.load(open(os.path.join(REPO, "tests", "golden", "index.json")))
    blobs = {}
    grads = {}
    for name in ("tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"):
        e = index[name]
        scene = synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{e['seed']:05d}_00", **e["kw"])
        for kind in ("segment", "instance", "semantic"):
            getattr(model_mod, f"export_{kind}_label").__defaults__ = (scene.num_points,)
        with tempfile.TemporaryDirectory() as wd:
            synthetic.write_reference_tree(wd, [scene])
            cwd = os.getcwd()
            os.chdir(wd)
            try:
                torch.manual_seed(1)
                net = model_mod.SegModel(exp_name="cap", cuda=False, sem_infer=False, ins_infer=False)
                net.load_state_dict(W.to_state_dict(wts, prefix=""), strict=False)
                net.epoch = "0"

                class Pinned(torch.nn.Module):
                    def forward(self, x):
                        return x * torch.from_numpy(cpu_ref.dropout_keep(x.shape[0]))
                net.classifier.dp1 = Pinned()
                seen = {}
                net.classifier.register_forward_hook(lambda m, i, o: seen.update(feat6=i[0].detach().numpy().copy(), logits=o.detach().numpy().copy()))
                out = net(torch.from_numpy(scene.data)[None], torch.from_numpy(scene.weak_label)[None], torch.tensor([[0]]))
                assert net.training
                # train.py:164-168 on one rank
                step_loss = torch.sum(out[0][:, 0]) / torch.sum(out[0][:, 1])
                step_loss.backward()
                for k, p in net.named_parameters():
                    grads[f"{name}.grad.{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
                for k, b in net.named_buffers():
                    grads[f"{name}.buf.{k}"] = b.detach().numpy().copy()
                grads[f"{name}.step_loss"] = np.array([float(step_loss)], dtype=np.float64)
            finally:
                os.chdir(cwd)
        loss = out[0].detach().numpy()
        blobs[f"{name}.loss"] = loss
        blobs[f"{name}.feat6"] = seen["feat6"]
        blobs[f"{name}.logits"] = seen["logits"]
        blobs[f"{name}.keep"] = cpu_ref.dropout_keep(seen["feat6"].shape[0])
        print(name, "loss", loss, "K", seen["feat6"].shape[0], flush=True)
        if "w.classifier.linear1.weight" not in blobs:
            for k, v in net.state_dict().items():
                if k.startswith("classifier.") and "running" not in k and "num_batches" not in k:
                    blobs["w." + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "train_tail.npz"), **blobs)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "train_grads.npz"), **grads)


if __name__ == "__main__":
    main()
