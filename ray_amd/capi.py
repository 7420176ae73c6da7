"""ctypes binding of include/ray_amd.h (libray_amd.so, built in-tree by __graft_entry__.build()).

This is plumbing only: structure layouts, prototypes and a loader that fails loudly when the HIP
library is missing -- there is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

PRIM_STRIDE = 9
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libray_amd.so")


class RtScene(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32),
        ("prim_kind", C.POINTER(C.c_int32)), ("prim_mat", C.POINTER(C.c_int32)),
        ("prim_outer", C.POINTER(C.c_int32)), ("prim_side", C.POINTER(C.c_int32)),
        ("prim_data", C.POINTER(C.c_double)),
        ("n_mats", C.c_int32),
        ("mat_kind", C.POINTER(C.c_int32)), ("mat_rgb", C.POINTER(C.c_float)),
        ("mat_fuzz", C.POINTER(C.c_float)), ("mat_ir", C.POINTER(C.c_double)),
        ("light_kind", C.c_int32),
        ("light_data", C.c_double * PRIM_STRIDE),
        ("background", C.c_float * 3),
    ]


class RtCamera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in
                ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + [("lens_radius", C.c_double)]

    def as_array(self) -> np.ndarray:
        return np.array([list(getattr(self, n)) for n, _ in self._fields_[:7]], dtype=np.float32)


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class SceneArrays:
    """Owns contiguous numpy copies of a flattened scene and the RtScene that points into them."""

    def __init__(self, flat: dict):
        self.flat = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in flat.items()}
        f = self.flat
        s = RtScene()
        s.n_prims = int(f["n_prims"])
        s.prim_kind = _ptr(f["prim_kind"], C.c_int32); s.prim_mat = _ptr(f["prim_mat"], C.c_int32)
        s.prim_outer = _ptr(f["prim_outer"], C.c_int32); s.prim_side = _ptr(f["prim_side"], C.c_int32)
        s.prim_data = _ptr(f["prim_data"], C.c_double)
        s.n_mats = int(f["n_mats"])
        s.mat_kind = _ptr(f["mat_kind"], C.c_int32); s.mat_rgb = _ptr(f["mat_rgb"], C.c_float)
        s.mat_fuzz = _ptr(f["mat_fuzz"], C.c_float); s.mat_ir = _ptr(f["mat_ir"], C.c_double)
        s.light_kind = int(f["light_kind"])
        for i in range(PRIM_STRIDE):
            s.light_data[i] = float(f["light_data"][i])
        for i in range(3):
            s.background[i] = float(f["background"][i])
        self.struct = s

    def ref(self):
        return C.byref(self.struct)


_PROTOS = {
    "rt_init": ([C.c_int], C.c_int),
    "rt_device_count": ([C.POINTER(C.c_int)], C.c_int),
    "rt_set_device": ([C.c_int], C.c_int),
    "rt_set_stream": ([C.c_void_p], C.c_int),
    "rt_reset_stream": ([], C.c_int),
    "rt_set_sample_slices": ([C.c_int, C.c_int], C.c_int),
    "rt_set_tail_limit": ([C.c_int64], C.c_int),
    "rt_shutdown": ([], None),
    "rt_last_error": ([], C.c_char_p),
    "rt_device_info": ([C.c_char_p, C.c_int], C.c_int),
    "rt_malloc": ([C.POINTER(C.c_void_p), C.c_uint64], C.c_int),
    "rt_free": ([C.c_void_p], C.c_int),
    "rt_set_shading_sort": ([C.c_int], C.c_int),
    "rt_set_tile_order": ([C.c_int], C.c_int),
    "rt_malloc_host": ([C.POINTER(C.c_void_p), C.c_uint64], C.c_int),
    "rt_free_host": ([C.c_void_p], C.c_int),
    "rt_memcpy_d2h": ([C.c_void_p, C.c_void_p, C.c_uint64], C.c_int),
    "rt_memcpy_h2d": ([C.c_void_p, C.c_void_p, C.c_uint64], C.c_int),
    "rt_memset": ([C.c_void_p, C.c_int, C.c_uint64], C.c_int),
    "rt_sync": ([], C.c_int),
    "rt_upload_scene": ([C.POINTER(RtScene)], C.c_int),
    "rt_camera_from_lookat": ([C.POINTER(C.c_float)] * 3 + [C.c_double] * 4 + [C.POINTER(RtCamera)], C.c_int),
    "rt_set_camera": ([C.POINTER(RtCamera)], C.c_int),
    "rt_primary_hits": ([C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p], C.c_int),
    "rt_hit_rays": ([C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 5, C.c_int),
    "rt_render": ([C.c_int] * 7 + [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p], C.c_int),
    "rt_render_noalloc": ([C.c_int] * 7 + [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p], C.c_int),
    "rt_render_bands": ([C.c_int] * 7 + [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p], C.c_int),
    "rt_render_pixels": ([C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p], C.c_int),
    "rt_trace_rays": ([C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p], C.c_int),
    "rt_tonemap": ([C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p], C.c_int),
    "rt_verify_mismatches": ([C.POINTER(C.c_uint64)], C.c_int),
    "rt_last_kernel_ms": ([C.POINTER(C.c_float), C.POINTER(C.c_int)], C.c_int),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen libray_amd.so and attach prototypes. Raises if the library was not built."""
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP library is required; there is no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own HIP / HSA runtime under the same sonames as /opt/rocm's. A process can
    # initialise the GPU through ONE copy only (the second finds "no HIP GPUs"), so when torch is installed it is
    # imported first and libray_amd.so then binds to the runtime that is already loaded. Callers that hand torch
    # tensors or streams to the library need this order anyway.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    for name, (args, res) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    return lib
