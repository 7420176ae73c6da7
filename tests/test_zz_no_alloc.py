"""rt_render_noalloc (include/ray_amd.h): the entry point that allocates nothing. Its scratch -- the tile
counter and the HBM part of the path stack -- is a file-scope device array of the library instead of
hipMalloc'ed buffers; results must equal rt_render's bit for bit. Two calls in a row, the second with a
smaller input (the array keeps the first call's contents), then the limits."""
import numpy as np
import pytest

import oracle_lib as O
from ray_amd import api, scenes

pytestmark = pytest.mark.gpu


def _noalloc(rt, W, H, spp, depth, seed, flags=0, y_begin=0, y_count=None, spp_begin=0, acc=None, rays=None):
    own = acc is None
    if own:
        acc, rays = rt.alloc(W * H * 16).zero(), rt.alloc(8).zero()
    rt._check(rt.lib.rt_render_noalloc(W, H, y_begin, H if y_count is None else y_count, spp_begin, spp, depth, seed, flags,
                                       acc.ptr, rays.ptr))
    out, n = acc.to_numpy(np.float32, (H, W, 4)), int(rays.to_numpy(np.uint64, (1,))[0])
    if own:
        acc.free(); rays.free()
    return out, n


def test_noalloc_equals_rt_render_twice_in_a_row(rt):
    sc = scenes.get_scene("random", half=6)       # glass and metal spheres: paths that reach depth 16
    rt.upload_scene(sc)
    # first call: a frame large enough for the full resident grid, depth 16 (10 stack levels in the scratch array)
    W, H, spp, depth = 640, 360, 4, 16
    want, want_rays = rt.render(W, H, spp, depth, seed=0xA110C)
    got, got_rays = _noalloc(rt, W, H, spp, depth, 0xA110C)
    assert got_rays == want_rays and got_rays > W * H * spp
    assert O.bits_equal(got, want)
    # second call, smaller input (fewer blocks, a shallower stack, a row band, accumulate on top of a first pass)
    W2, H2 = 97, 41
    want2 = rt.alloc(W2 * H2 * 16).zero(); r2 = rt.alloc(8).zero()
    rt.render_into(W2, H2, want2, 2, 9, seed=7, y_begin=3, y_count=30, rays=r2)
    rt.render_into(W2, H2, want2, 3, 9, seed=7, y_begin=3, y_count=30, spp_begin=2, flags=api.RENDER_ACCUMULATE, rays=r2)
    acc = rt.alloc(W2 * H2 * 16).zero(); r = rt.alloc(8).zero()
    _noalloc(rt, W2, H2, 2, 9, 7, y_begin=3, y_count=30, acc=acc, rays=r)
    got2, n2 = _noalloc(rt, W2, H2, 3, 9, 7, flags=api.RENDER_ACCUMULATE, y_begin=3, y_count=30, spp_begin=2, acc=acc, rays=r)
    assert n2 == int(r2.to_numpy(np.uint64, (1,))[0])
    assert O.bits_equal(got2, want2.to_numpy(np.float32, (H2, W2, 4)))
    assert (got2[:3] == 0).all() and (got2[33:] == 0).all() and (got2[3:33, :, 3] == 5).all()
    # and against the oracle, so that "equal" is not "equally wrong"
    oacc = np.zeros((H2, W2, 4), np.float32)
    O.render(sc, W2, H2, 2, 9, rng_mode=1, seed64=7, math_mode=1, y_begin=3, y_count=30, accum=oacc, n_threads=8)
    first, _ = _noalloc(rt, W2, H2, 2, 9, 7, y_begin=3, y_count=30)
    assert O.same_values(first, oacc)
    # the first (large) case once more after the small one
    again, again_rays = _noalloc(rt, W, H, spp, depth, 0xA110C)
    assert again_rays == want_rays and O.bits_equal(again, want)
    for b in (want2, r2, acc, r):
        b.free()


def test_noalloc_limits(rt):
    rt.upload_scene(scenes.builtin_scene())
    acc, rays = rt.alloc(16 * 16 * 16).zero(), rt.alloc(8).zero()
    assert rt.lib.rt_render_noalloc(16, 16, 0, 16, 0, 1, 17, 0, 0, acc.ptr, rays.ptr) != 0
    assert b"exceeds the limit of 16" in rt.lib.rt_last_error()
    assert rt.lib.rt_render_noalloc(16, 16, 0, 16, 0, 1, 8, 0, api.RENDER_VERIFY, acc.ptr, rays.ptr) != 0
    assert rt.lib.rt_render_noalloc(16, 16, 0, 16, 0, 1, 8, 0, 0, None, rays.ptr) != 0
    # black path (depth 0) and an empty band need no scratch at all
    assert rt.lib.rt_render_noalloc(16, 16, 0, 16, 0, 3, 0, 0, 0, acc.ptr, rays.ptr) == 0
    a = acc.to_numpy(np.float32, (16, 16, 4))
    assert (a[..., 3] == 3).all() and (a[..., :3] == 0).all()
    assert rt.lib.rt_render_noalloc(16, 16, 5, 0, 0, 3, 8, 0, 0, acc.ptr, rays.ptr) == 0
    acc.free(); rays.free()
