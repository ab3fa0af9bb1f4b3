"""The torch-free reader of the reference's one-tensor `.pth` files (seggroup_amd/pth.py) against torch.load."""
import numpy as np
import pytest


def test_pth_reader_equals_torch_load(tmp_path):
    import torch
    from seggroup_amd import pth
    cases = [torch.arange(24, dtype=torch.float32).reshape(4, 6), torch.arange(24).reshape(6, 4)[:, 1:3],      # a strided view
             torch.tensor([5]), torch.arange(10)[::2], torch.zeros(0, 2, dtype=torch.int64),
             torch.randint(-1, 40, (1000, 2)), torch.rand(1000, 6), torch.arange(6, dtype=torch.int32).reshape(2, 3).t()]
    for i, t in enumerate(cases):
        p = str(tmp_path / f"t{i}.pth")
        torch.save(t, p)
        a = pth.load_tensor(p)
        assert a.dtype == t.numpy().dtype and a.shape == tuple(t.shape) and np.array_equal(a, t.numpy()) and a.flags["C_CONTIGUOUS"]


def test_pth_reader_refuses_anything_but_a_plain_tensor(tmp_path):
    import torch
    from seggroup_amd import pth
    p = str(tmp_path / "d.pth")
    torch.save({"state_dict": {"w": torch.zeros(3)}}, p)
    with pytest.raises(Exception):
        pth.load_tensor(p)
    q = str(tmp_path / "x.pth")
    open(q, "wb").write(b"not a zip")
    with pytest.raises(Exception):
        pth.load_tensor(q)


def test_pth_reader_refuses_a_view_beyond_its_storage():
    from seggroup_amd import pth
    store = (np.arange(12, dtype=np.float32).tobytes(), np.float32)
    assert pth._rebuild_tensor_v2(store, 2, (2, 5), (5, 1)).shape == (2, 5)          # last element = flat[11]
    for off, size, stride in ((3, (2, 5), (5, 1)), (0, (4, 4), (4, 1)), (-1, (2,), (1,)), (0, (2,), (-1,)), (12, (1,), (1,)), (0, (2, 2), (1,))):
        with pytest.raises(ValueError):
            pth._rebuild_tensor_v2(store, off, size, stride)
    assert pth._rebuild_tensor_v2(store, 12, (0, 3), (3, 1)).shape == (0, 3)
