"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, its host-only arithmetic (camera::resetCam) matches the reference golden, and without a GPU
the compute entry points fail loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from ray_amd import capi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return capi.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ray_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ray_amd.h but not exported"
    assert declared == set(capi.EXPORTED_SYMBOLS)


def test_camera_from_lookat_matches_reference_golden(lib):
    g = O.golden("camera")
    vecs, prm = g["cfg_vecs"], g["cfg_params"]
    for i in range(vecs.shape[0]):
        cam = capi.RtCamera()
        v = [np.ascontiguousarray(vecs[i, k]) for k in range(3)]
        fp = [a.ctypes.data_as(C.POINTER(C.c_float)) for a in v]
        rc = lib.rt_camera_from_lookat(fp[0], fp[1], fp[2], *[float(x) for x in prm[i]], C.byref(cam))
        assert rc == 0
        assert O.bits_equal(cam.as_array(), g[f"cam{i}_frame"])
        assert cam.lens_radius == g[f"cam{i}_lens_radius"][0]


def test_no_cpu_fallback_without_a_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.rt_init(0) != 0
    assert b"no CPU fallback" in lib.rt_last_error()
    sa = capi.SceneArrays(scenes.builtin_scene().flatten())
    assert lib.rt_upload_scene(sa.ref()) != 0
    assert lib.rt_render(8, 8, 0, 8, 0, 1, 4, 0, 0, None, None) != 0
    assert lib.rt_render_bands(8, 8, 0, 2, 0, 1, 4, 0, 0, None, None) != 0
    assert lib.rt_render_pixels(8, 8, 1, None, 0, 1, 4, 0, 0, None, None) != 0
    assert b"rt_init" in lib.rt_last_error()


def test_scene_flattening_box_order_and_text_roundtrip():
    sc = scenes.box_scene()
    flat = sc.flatten()
    assert flat["n_prims"] == 13 + 26 * 6
    i = int(np.where(flat["prim_side"] == 0)[0][0])
    # shapes.h:347-354: sides are xy(k=p1.z), xy(k=p0.z), xz(k=p1.y), xz(k=p0.y), yz(k=p1.x), yz(k=p0.x)
    assert list(flat["prim_kind"][i:i + 6]) == [1, 1, 2, 2, 3, 3]
    assert list(flat["prim_side"][i:i + 6]) == [0, 1, 2, 3, 4, 5]
    assert len(set(flat["prim_outer"][i:i + 6])) == 1
    txt = sc.to_text()
    assert txt.count("\nbox ") == 26 and "light xz_rect" in txt
    assert scenes.random_scene(11).flatten()["n_prims"] in range(470, 490)
    assert scenes.tri_mesh_scene().flatten()["n_prims"] == 9800 + 2


def test_two_context_objects_each_reach_their_own_device():
    """The C library keeps one context per GPU and acts on the calling thread's CURRENT one, so api.Context rebinds its device before
    every call. Checked here without a GPU: a stand-in library that records (current device, call) pairs must see every call
    of each object arrive with that object's device current, however the calls of two objects interleave, close() included."""
    from ray_amd import api

    class FakeLib:
        def __init__(self):
            self.current, self.log = None, []

        def rt_set_device(self, d):
            self.current = d
            return 0

        def rt_last_error(self):
            return b""

        def __getattr__(self, name):
            def fn(*args):
                self.log.append((self.current, name))
                return 0
            return fn

    fake = FakeLib()
    a, b = api.Context.__new__(api.Context), api.Context.__new__(api.Context)
    for obj, dev in ((a, 0), (b, 1)):
        obj.lib, obj.device, obj._scene_keep = api._BoundLib(fake, dev), dev, None
    a.sync(); b.sync(); a.set_sample_slices(4, 2); b.set_tail_limit(7); a.last_kernel_ms(); b.last_kernel_ms()
    b.close(); a.sync(); a.close()
    assert fake.log == [(0, "rt_sync"), (1, "rt_sync"), (0, "rt_set_sample_slices"), (1, "rt_set_tail_limit"), (0, "rt_last_kernel_ms"),
                        (1, "rt_last_kernel_ms"), (1, "rt_shutdown"), (0, "rt_sync"), (0, "rt_shutdown")]
