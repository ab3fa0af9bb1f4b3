"""The native pack loader (csrc/loader.cpp, sg_loader_*) on the GPU: packs with the int32 adjacency of round 4 and packs with the reference's
int64 rows must both come out as the same sg_scene (d_adj int64 either way), equal to the arrays that were staged."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_fixture_scene

pytestmark = pytest.mark.gpu


def _from_device(ptr, shape, dtype):
    """copy a device array owned by a loader slot back to the host through the process's HIP runtime"""
    from seggroup_amd import hip
    rt = hip._load_hip_runtime()
    out = np.empty(shape, dtype)
    addr = C.cast(ptr, C.c_void_p).value
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert rt.hipMemcpy(out.ctypes.data, addr, out.nbytes, 2) == 0           # 2 = hipMemcpyDeviceToHost
    return out


def test_loader_serves_int32_and_int64_adjacency_packs_alike(tmp_path, golden_index):
    from seggroup_amd import cache

    scene = make_fixture_scene(golden_index, "tiny_4k")
    want = cache.stage_arrays(scene.data, scene.weak_label, scene.seg, scene.adj, scene.unmap, scene.gt)
    p32, p64 = str(tmp_path / "a32.sgpack"), str(tmp_path / "a64.sgpack")
    cache.write_pack(p32, "a32", want)
    cache.write_pack(p64, "a64", want, adj_int32=False)
    assert os.path.getsize(p64) - os.path.getsize(p32) >= want["adj"].shape[0] * 8 - 64          # half the adjacency's bytes
    assert np.array_equal(cache.read_pack(p32)["adj"], want["adj"]) and np.array_equal(cache.read_pack(p64)["adj"], want["adj"])
    ld = cache.PackLoader(threads=2, slots=4, slot_bytes=os.path.getsize(p64), device="cuda:0")
    try:
        got = [ld.wait(ld.submit(p)) for p in (p32, p64, p32)]
        for ls, nm in zip(got, ("a32", "a64", "a32")):
            assert ls.name == nm
            assert (ls.N, ls.S, ls.E0, ls.V) == (want["data"].shape[0], want["seg_first"].shape[0], want["adj"].shape[0], want["unmap"].shape[0])
            c = ls.c_struct
            assert np.array_equal(_from_device(c.d_adj, (ls.E0, 2), np.int64), want["adj"])      # int64 rows either way
            assert np.array_equal(_from_device(c.d_data, (ls.N, 6), np.float32), want["data"])
            assert np.array_equal(_from_device(c.d_unmap, (ls.V,), np.int32), want["unmap"])
            assert np.array_equal(ls.h_seg_size, want["seg_size"])
        for ls in got:
            ls.release()
    finally:
        ld.close()
