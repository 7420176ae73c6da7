"""Python host wrapper over the C ABI (include/ray_amd.h). Thin by design: it allocates device
buffers, moves numpy arrays across and calls the library; all ray work runs in the HIP kernels.
Used by tests/, bench.py and __graft_entry__.smoke(). There is no CPU path here."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import capi
from .capi import RtCamera, SceneArrays

RENDER_ACCUMULATE = 1
RENDER_NORMALS = 2
RENDER_VERIFY = 4
RENDER_NOWAIT = 8
RENDER_CONTINUE = 16


class RtError(RuntimeError):
    pass


class DeviceBuffer:
    """Device memory obtained from rt_malloc (freed on close / garbage collection)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._check(ctx.lib.rt_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def zero(self):
        self.ctx._check(self.ctx.lib.rt_memset(self.ptr, 0, self.nbytes))
        return self

    def to_numpy(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.rt_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def from_numpy(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.rt_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes))
        return self

    def free(self):
        if self.ptr and self.ctx.lib is not None:
            self.ctx.lib.rt_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _dev_ptr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):       # torch tensor on the GPU
        return int(x.data_ptr())
    return int(x)


class _BoundLib:
    """The C library as one Context object sees it. The library keeps one context per GPU and every entry point acts on the
    calling thread's CURRENT one (include/ray_amd.h), so with two Context objects in a process each call must first make
    its own object's context current: every function fetched from here does rt_set_device(device) before the call."""

    _UNBOUND = ("rt_init", "rt_set_device", "rt_last_error", "rt_device_count", "rt_camera_from_lookat")

    def __init__(self, lib, device: int):
        self._lib, self._device = lib, device

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self._UNBOUND:
            return fn
        lib, device = self._lib, self._device

        def bound(*args):
            if lib.rt_set_device(device) != 0:
                raise RtError(lib.rt_last_error().decode())
            return fn(*args)
        bound.__name__ = name
        setattr(self, name, bound)      # (argtypes of diagnostic entries are set on the raw library: see raw())
        return bound

    def raw(self):
        return self._lib


class Context:
    def __init__(self, device: int = 0, lib_path: str = capi.LIB_PATH):
        raw = capi.load_library(lib_path)
        if raw.rt_init(device) != 0:
            raise RtError(raw.rt_last_error().decode())
        self.lib = _BoundLib(raw, device)
        self.device = device
        self._scene_keep = None

    def _check(self, rc: int):
        if rc != 0:
            raise RtError(self.lib.rt_last_error().decode())

    def close(self):
        if self.lib is not None:
            self.lib.rt_shutdown()
            self.lib = None

    def device_info(self) -> dict:
        buf = C.create_string_buffer(256)
        self._check(self.lib.rt_device_info(buf, 256))
        name, arch, cus, mhz = buf.value.decode().split("|")
        return dict(name=name, arch=arch, cus=int(cus), clock_mhz=int(mhz))

    def set_stream(self, stream=None):
        """rt_set_stream: run the kernels on the caller's HIP stream (a torch.cuda.Stream, a raw hipStream_t, or None
        for the library's own), so that they are ordered with the caller's asynchronous work on that stream."""
        if stream is None:
            self._check(self.lib.rt_reset_stream())
        else:
            self._check(self.lib.rt_set_stream(int(getattr(stream, "cuda_stream", stream)) or None))   # 0 = HIP's default stream

    def sync(self):
        self._check(self.lib.rt_sync())

    def set_sample_slices(self, min_samples: int = 8, max_slices: int = 16):
        """rt_set_sample_slices: how a pixel's samples are cut into units of work (0 = whole pixels); same bits either way."""
        self._check(self.lib.rt_set_sample_slices(min_samples, max_slices))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def set_tail_limit(self, megabytes: int = 1400):
        """rt_set_tail_limit: upper limit of the colour buffer for the end of a frame (0 = none); same bits either way."""
        self._check(self.lib.rt_set_tail_limit(megabytes))

    def set_tile_order(self, mode: int = 0):
        """rt_set_tile_order: 1 = frames of 32 samples and more hand their tiles out in order of cost (measured by their first sample); same bits."""
        self._check(self.lib.rt_set_tile_order(mode))

    def set_shading_sort(self, mode: int = 0):
        """rt_set_shading_sort: 1 = block-sorted shading where its kernel exists (an experiment, measured slower; 0 = default); same bits."""
        self._check(self.lib.rt_set_shading_sort(mode))

    # ---- scene / camera --------------------------------------------------------------------------
    def upload_scene(self, scene, set_camera: bool = True):
        sa = scene if isinstance(scene, SceneArrays) else SceneArrays(scene.flatten())
        self._check(self.lib.rt_upload_scene(sa.ref()))
        self._scene_keep = sa
        if set_camera and hasattr(scene, "camera"):
            self.set_camera(self.camera_from(scene.camera))
        return sa

    def camera_from(self, cam) -> RtCamera:
        out = RtCamera()
        v = [np.array(a, dtype=np.float32) for a in (cam.lookfrom, cam.lookat, cam.vup)]
        fp = [a.ctypes.data_as(C.POINTER(C.c_float)) for a in v]
        self._check(self.lib.rt_camera_from_lookat(fp[0], fp[1], fp[2], cam.focus, cam.vfov, cam.aspect, cam.aperture,
                                                   C.byref(out)))
        return out

    def set_camera(self, cam: RtCamera):
        self._check(self.lib.rt_set_camera(C.byref(cam)))

    # ---- hot path --------------------------------------------------------------------------------
    def primary_hits(self, W: int, H: int, seed: int = 0):
        dp, dt = self.alloc(W * H * 4), self.alloc(W * H * 8)
        self._check(self.lib.rt_primary_hits(W, H, seed, dp.ptr, dt.ptr))
        prim, t = dp.to_numpy(np.int32, (H, W)), dt.to_numpy(np.float64, (H, W))
        dp.free(); dt.free()
        return prim, t

    def hit_rays(self, ro: np.ndarray, rd: np.ndarray, t_min: float, t_max: float):
        ro = np.ascontiguousarray(ro, dtype=np.float32); rd = np.ascontiguousarray(rd, dtype=np.float32)
        n = ro.shape[0]
        d_ro, d_rd = self.alloc(max(ro.nbytes, 4)).from_numpy(ro), self.alloc(max(rd.nbytes, 4)).from_numpy(rd)
        d_prim, d_t = self.alloc(max(n * 4, 4)), self.alloc(max(n * 8, 8))
        d_p, d_n, d_ff = self.alloc(max(n * 12, 4)), self.alloc(max(n * 12, 4)), self.alloc(max(n, 1))
        self._check(self.lib.rt_hit_rays(n, d_ro.ptr, d_rd.ptr, t_min, t_max, d_prim.ptr, d_t.ptr, d_p.ptr, d_n.ptr, d_ff.ptr))
        out = (d_prim.to_numpy(np.int32, (n,)), d_t.to_numpy(np.float64, (n,)), d_p.to_numpy(np.float32, (n, 3)),
               d_n.to_numpy(np.float32, (n, 3)), d_ff.to_numpy(np.uint8, (n,)))
        for b in (d_ro, d_rd, d_prim, d_t, d_p, d_n, d_ff):
            b.free()
        return out

    def trace_rays(self, ro: np.ndarray, rd: np.ndarray, depth: int, seed: int = 0, flags: int = 0) -> np.ndarray:
        ro = np.ascontiguousarray(ro, dtype=np.float32); rd = np.ascontiguousarray(rd, dtype=np.float32)
        n = ro.shape[0]
        d_ro, d_rd = self.alloc(max(ro.nbytes, 4)).from_numpy(ro), self.alloc(max(rd.nbytes, 4)).from_numpy(rd)
        d_rgb = self.alloc(max(n * 12, 4))
        self._check(self.lib.rt_trace_rays(n, d_ro.ptr, d_rd.ptr, depth, seed, flags, d_rgb.ptr))
        out = d_rgb.to_numpy(np.float32, (n, 3))
        for b in (d_ro, d_rd, d_rgb):
            b.free()
        return out

    def render_into(self, W, H, accum, spp, depth, seed=0, flags=0, y_begin=0, y_count=None, spp_begin=0, rays=None,
                    band_first=0, band_stride=1):
        """rt_render into a caller-owned device accumulator ([H][W][4] float32: DeviceBuffer, torch
        tensor or raw pointer). `rays`: optional device uint64 counter. band_stride > 1: rt_render_bands
        (8-row bands band_first, band_first + band_stride, ... of the whole frame)."""
        if band_stride != 1 or band_first != 0:
            self._check(self.lib.rt_render_bands(W, H, band_first, band_stride, spp_begin, spp, depth, seed, flags,
                                                 _dev_ptr(accum), _dev_ptr(rays)))
            return
        self._check(self.lib.rt_render(W, H, y_begin, H if y_count is None else y_count, spp_begin, spp, depth, seed, flags,
                                       _dev_ptr(accum), _dev_ptr(rays)))

    def render_pixels(self, W, H, accum, pixels, spp, depth, seed=0, flags=0, spp_begin=0, rays=None):
        """rt_render_pixels: `pixels` = int32 pixel indices y * W + x (numpy array, or a device buffer / torch tensor
        with `n` given by its length) rendered into the caller-owned device accumulator."""
        own = None
        if isinstance(pixels, np.ndarray):
            px = np.ascontiguousarray(pixels, dtype=np.int32)
            n = int(px.size)
            own = self.alloc(max(px.nbytes, 4)).from_numpy(px)
            pixels = own
        else:
            n = int(pixels.numel()) if hasattr(pixels, "numel") else int(pixels.nbytes // 4)
        try:
            self._check(self.lib.rt_render_pixels(W, H, n, _dev_ptr(pixels), spp_begin, spp, depth, seed, flags, _dev_ptr(accum), _dev_ptr(rays)))
        finally:
            if own is not None:
                own.free()

    def render(self, W, H, spp, depth, seed=0, flags=0, y_begin=0, y_count=None, spp_begin=0, band_first=0, band_stride=1):
        """Convenience: render into a fresh zeroed accumulator and return (accum[H,W,4], n_rays)."""
        acc = self.alloc(W * H * 16).zero()
        rays = self.alloc(8).zero()
        self.render_into(W, H, acc, spp, depth, seed, flags, y_begin, y_count, spp_begin, rays, band_first, band_stride)
        out = acc.to_numpy(np.float32, (H, W, 4))
        n = int(rays.to_numpy(np.uint64, (1,))[0])
        acc.free(); rays.free()
        return out, n

    def tonemap(self, W, H, accum, n_samples: int = 0) -> np.ndarray:
        own = None
        if isinstance(accum, np.ndarray):
            own = self.alloc(accum.nbytes).from_numpy(accum.astype(np.float32))
            accum = own
        img = self.alloc(W * H * 4)
        self._check(self.lib.rt_tonemap(W, H, _dev_ptr(accum), n_samples, img.ptr))
        out = img.to_numpy(np.uint32, (H, W))
        img.free()
        if own is not None:
            own.free()
        return out

    def verify_mismatches(self) -> int:
        n = C.c_uint64(0)
        self._check(self.lib.rt_verify_mismatches(C.byref(n)))
        return n.value

    def last_kernel_ms(self):
        ms, n = C.c_float(0), C.c_int(0)
        self.lib.rt_last_kernel_ms(C.byref(ms), C.byref(n))
        return ms.value, n.value


class Progressive:
    """RayView::go (rayview.cpp:98-117) for the GPU: each go() is one time slice of full-frame passes that
    CONTINUE one device accumulator in sample order (RT_RENDER_CONTINUE: after n samples it holds what one call of n
    samples leaves, bit for bit); sample indices continue across slices, reset() (scene or camera changed) starts
    again from sample 0. Mirrors ProgressiveView of ray_amd/host/ray_amd_host.h."""

    def __init__(self, ctx: Context, W: int, H: int, depth: int, seed: int = 0):
        self.ctx, self.W, self.H, self.depth, self.seed = ctx, W, H, depth, seed
        self.acc = ctx.alloc(W * H * 16)
        self.done, self.per_pass = 0, 1

    def reset(self, scene=None):
        if scene is not None:
            self.ctx.upload_scene(scene)
        self.done = 0

    def go(self, budget_ms: float = 16.0) -> int:
        used = 0.0
        while True:
            flags = RENDER_CONTINUE if self.done else 0
            self.ctx.render_into(self.W, self.H, self.acc, self.per_pass, self.depth, seed=self.seed, flags=flags, spp_begin=self.done)
            ms, _ = self.ctx.last_kernel_ms()
            self.done += self.per_pass
            used += ms
            per_sample = ms / self.per_pass
            nxt = int(0.5 * budget_ms / per_sample) if per_sample > 0 else self.per_pass * 2
            self.per_pass = min(max(nxt, 1), 64)
            if used + ms > budget_ms:
                return self.done

    def accumulators(self) -> np.ndarray:
        return self.acc.to_numpy(np.float32, (self.H, self.W, 4))

    def image(self) -> np.ndarray:
        return self.ctx.tonemap(self.W, self.H, self.acc)
