"""GPU parity tests: the HIP path, called through the C ABI (libray_amd.so), against the CPU oracle
on the same seeded inputs and against the committed golden fixtures (which come from the reference's
own code). Bit-exact everywhere: integer ids, double t, and FP32 radiance sums."""
import numpy as np
import pytest

import oracle_lib as O
from oracle.make_golden import golden_scenes
from ray_amd import scenes

pytestmark = pytest.mark.gpu


def _outer_side(sc, prim):
    flat = sc.flatten()
    outer = np.where(prim >= 0, flat["prim_outer"][np.maximum(prim, 0)], -1)
    side = np.where(prim >= 0, flat["prim_side"][np.maximum(prim, 0)], -1)
    return outer, side


# ---- primary-hit ids vs the reference's own output (golden) -------------------------------------------
@pytest.mark.parametrize("sname,W,H", [("builtin_a0", 640, 360), ("random11_a0", 640, 360), ("box_a0", 640, 360),
                                       ("builtin_a0", 1920, 1080),
                                       ("random11_a0", 3840, 2160)])   # north_star's frame: 485 spheres at 4K, ids from the reference itself
def test_primary_ids_match_reference_golden(rt, sname, W, H):
    import hashlib
    g = O.golden(f"primary_{sname}_{W}x{H}")
    sc = golden_scenes()[sname]
    rt.upload_scene(sc)
    prim, t = rt.primary_hits(W, H, seed=0)
    outer, side = _outer_side(sc, prim)
    assert np.array_equal(outer, g["prim_id"])
    assert np.array_equal(side, g["box_side"])
    assert O.bits_equal(t[::9, ::9].copy(), g["t_sub"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(t.tobytes()).digest(), np.uint8), g["t_sha256"])


# ---- primary hits with a finite aperture (lens samples from the counter RNG) vs the oracle ------------
@pytest.mark.parametrize("name,kw,W,H", [("builtin", {}, 320, 180), ("random", {"half": 11}, 320, 180),
                                         ("box", {"aperture": 0.05}, 160, 90), ("tris", {"n": 21}, 160, 90),
                                         ("mixed", {"half": 3, "n": 11}, 160, 90), ("smoke", {"aperture": 0.1}, 160, 90)])
def test_primary_hits_match_oracle(rt, name, kw, W, H):
    sc = scenes.get_scene(name, **kw)
    rt.upload_scene(sc)
    prim, t = rt.primary_hits(W, H, seed=0xABCDEF0123)
    oprim, ot = O.primary(sc, W, H, rng_mode=1, seed=0xABCDEF0123)
    assert np.array_equal(prim, oprim)
    assert O.bits_equal(t, ot)
    assert (prim >= 0).mean() > 0.3


# ---- hitFromList for arbitrary rays, including the edge cases of the shape fixtures --------------------
def test_hit_rays_match_oracle_and_edge_cases(rt):
    sc = scenes.get_scene("mixed", half=3, n=9)
    rt.upload_scene(sc)
    rng = np.random.default_rng(1)
    n = 20000
    ro = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(12.0)
    ro[:, 1] = np.abs(ro[:, 1]) + np.float32(0.05)
    rd = (rng.random((n, 3), dtype=np.float32) - 0.5)
    rd[::7, 1] = 0.0            # rays parallel to the ground plane
    rd[::11] = 0.0              # zero direction: a = 0 -> inf/NaN roots
    rd[5::13, 0] = np.float32(np.nan)
    for t_min, t_max in [(0.001, np.inf), (0.5, 6.0)]:
        got = rt.hit_rays(ro, rd, t_min, t_max)
        ref = O.hit_rays(sc, ro, rd, t_min, t_max)
        assert np.array_equal(got[0], ref[0])
        for a, b in zip(got[1:], ref[1:]):
            assert O.same_bits_or_nan(a, b)
    assert rt.hit_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0.001, np.inf)[0].shape == (0,)


def test_trace_rays_match_oracle(rt):
    """calc::ray_color for caller-supplied rays (the single-ray entry point of calc.h:89-94)."""
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    rng = np.random.default_rng(2)
    n = 5000
    ro = np.tile(np.array([0, 0, 1], np.float32), (n, 1)) + rng.standard_normal((n, 3)).astype(np.float32) * np.float32(0.05)
    rd = np.tile(np.array([0, 0, -2], np.float32), (n, 1)) + rng.standard_normal((n, 3)).astype(np.float32) * np.float32(0.8)
    for depth, flags in [(10, 0), (1, 0), (5, 2)]:
        got = rt.trace_rays(ro, rd, depth, seed=77, flags=flags)
        ref = O.trace_rays(sc, ro, rd, depth, seed=77, flags=flags)
        assert O.same_bits_or_nan(got, ref)


def test_shape_fixture_rays_through_gpu(rt):
    """Every sphere / rect case of shapehits.npz (tangent, inside, on-surface, parallel, edge) as a
    one-primitive scene batch: group the fixture by primitive, trace its rays, compare with the
    reference's recorded hit_record."""
    g = O.golden("shapehits")
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    cam = scenes.rayview_camera()
    # spheres: 200 distinct primitives x their rays (t_max = inf cases only: t_max is per call)
    sel = np.where(np.isinf(g["sph_tmax"]))[0][:3000]
    for i in sel[:300]:
        c = g["sph_prm"][i]
        sc = scenes.Scene(mats, [scenes.Shape("sphere", tuple(float(v) for v in c), 0)], None, cam)
        rt.upload_scene(sc, set_camera=False)
        prim, t, p, nrm, ff = rt.hit_rays(g["sph_ray_o"][i:i + 1], g["sph_ray_d"][i:i + 1], 0.001, np.inf)
        assert (prim[0] >= 0) == bool(g["sph_has"][i])
        if prim[0] >= 0:
            assert O.same_bits_or_nan(t[0], g["sph_t"][i]) and O.same_bits_or_nan(p[0], g["sph_p"][i])
            assert O.same_bits_or_nan(nrm[0], g["sph_n"][i]) and ff[0] == g["sph_ff"][i]
    sel = np.where(np.isinf(g["rect_tmax"]))[0]
    kinds = ["xy_rect", "xz_rect", "yz_rect"]
    seen_special = 0
    for i in list(sel[:200]) + [j for j in sel if not np.isfinite(g["rect_t"][j])][:50]:
        q = g["rect_prm"][i]
        sc = scenes.Scene(mats, [scenes.Shape(kinds[int(g["rect_axis"][i])], tuple(float(v) for v in q), 0)], None, cam)
        rt.upload_scene(sc, set_camera=False)
        prim, t, p, nrm, ff = rt.hit_rays(g["rect_ray_o"][i:i + 1], g["rect_ray_d"][i:i + 1], 0.001, np.inf)
        assert (prim[0] >= 0) == bool(g["rect_has"][i]), i
        if prim[0] >= 0:
            seen_special += not np.isfinite(g["rect_t"][i])
            assert O.same_bits_or_nan(t[0], g["rect_t"][i]) and O.same_bits_or_nan(p[0], g["rect_p"][i])
            assert O.same_bits_or_nan(nrm[0], g["rect_n"][i]) and ff[0] == g["rect_ff"][i]


# ---- calc::ray_color: per-pixel radiance sums, bit-exact vs the oracle on the same RNG streams ---------
@pytest.mark.parametrize("name,kw,W,H,spp,depth", [
    ("builtin", {}, 160, 90, 4, 8),
    ("builtin", {}, 64, 36, 16, 50),
    ("random", {"half": 3}, 96, 54, 4, 10),
    ("random", {"half": 11}, 64, 36, 2, 8),
    ("box", {"aperture": 0.05}, 64, 36, 4, 6),
    ("box", {"aperture": 0.05, "light_on_panel": True}, 32, 18, 4, 6),
    ("tris", {"n": 15}, 64, 36, 4, 6),
    ("mixed", {"half": 2, "n": 9}, 64, 36, 4, 8),
    ("smoke", {}, 96, 54, 4, 10),                   # constant_medium over a box and over spheres + isotropic (shapes.h:374-455)
    ("smoke", {"aperture": 0.2, "density": 2.5}, 64, 36, 4, 16),
    ("builtin", {"light": "xy"}, 96, 54, 4, 8),     # lights = the scene's XY `rectangle`: shapes.h:202-219 incl. quirk Q2
    ("builtin", {"light": "yz"}, 96, 54, 4, 8),     # lights = a yz_rect: base-class pdf_value / random (RT_LIGHT_BASE)
])
def test_radiance_sums_bit_exact_vs_oracle(rt, name, kw, W, H, spp, depth):
    sc = scenes.get_scene(name, **kw)
    rt.upload_scene(sc)
    acc, rays = rt.render(W, H, spp, depth, seed=0xC0FFEE)
    oacc, orays = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=0xC0FFEE, math_mode=1, n_threads=8)
    assert rays == orays
    assert O.same_bits_or_nan(acc, oacc)            # signed zeros included; NaN payloads are not part of the contract
    if kw.get("light_on_panel"):
        assert np.isnan(oacc).any()
    if kw.get("light"):
        assert sc.flatten()["light_kind"] == {"xy": scenes.PRIM_XY_RECT, "yz": scenes.LIGHT_BASE}[kw["light"]]
        assert np.array_equal(np.isnan(acc), np.isnan(oacc)) and np.isfinite(oacc[..., :3]).mean() > 0.2


@pytest.mark.parametrize("name,kw,W,H,spp,depth", [
    ("random", {"half": 11}, 960, 540, 4, 8),
    ("builtin", {}, 640, 360, 8, 10),
    ("mixed", {"half": 5, "n": 9}, 320, 180, 2, 8),
    ("tris", {"n": 71}, 480, 270, 2, 8),          # 9800 triangles: 10 chunks of the triangle group
    # the shared level 0 of the triangle filter: the mixed scene, where a few lanes of a wave need a chunk (entries
    # shared by several lanes, remainders kept in the queue), on a ragged frame (lanes outside the frame are not in the scan)
    ("mixed", {"half": 11, "n": 71}, 1283, 721, 1, 8),
    ("tris", {"n": 40}, 333, 187, 2, 8),
])
def test_fast_scan_equals_list_order_scan_at_scale(rt, name, kw, W, H, spp, depth):
    """RT_RENDER_VERIFY: every closest-hit scan of the frame (millions of primary and bounce rays) is
    repeated on the device as the reference's list-order loop (hitPosition.cpp:6-22); the sphere
    filter + exact path of rt_scan.hpp must give the same primitive and the same t bits every time,
    and the verified frame must equal the plain one."""
    sc = scenes.get_scene(name, **kw)
    rt.upload_scene(sc)
    acc, rays = rt.render(W, H, spp, depth, seed=0xBEEF)
    acc_v, rays_v = rt.render(W, H, spp, depth, seed=0xBEEF, flags=4)
    assert rt.verify_mismatches() == 0
    assert rays == rays_v and rays > W * H * spp
    assert O.same_bits_or_nan(acc, acc_v)


def test_render_edge_cases(rt):
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    # depth 0: no ray is traced, every sample is black (calc.cpp:96-97)
    acc, rays = rt.render(32, 18, 3, 0, seed=1)
    assert rays == 0 and (acc[..., :3] == 0).all() and (acc[..., 3] == 3).all()
    # zero samples, ragged sizes (not multiples of the 16x16 tile), single pixel
    acc, rays = rt.render(33, 17, 0, 8, seed=1)
    assert rays == 0 and (acc == 0).all()
    for W, H in [(1, 1), (17, 5), (250, 3)]:
        acc, rays = rt.render(W, H, 2, 8, seed=5)
        oacc, orays = O.render(sc, W, H, 2, 8, rng_mode=1, seed64=5, math_mode=1)
        assert rays == orays and O.same_bits_or_nan(acc, oacc)
    # normal-visualisation mode (calc.cpp:105-109)
    acc, _ = rt.render(64, 36, 2, 8, seed=5, flags=2)
    oacc, _ = O.render(sc, 64, 36, 2, 8, rng_mode=1, seed64=5, math_mode=1, flags=2)
    assert O.same_bits_or_nan(acc, oacc)
    # empty world: every ray returns the background
    empty = scenes.Scene(sc.mats, [], sc.light, sc.camera)
    rt.upload_scene(empty)
    acc, rays = rt.render(16, 16, 2, 8, seed=5)
    assert rays == 16 * 16 * 2 and np.allclose(acc[..., :3], 2 * np.array([0.5, 0.7, 1.0], np.float32))
    prim, _ = rt.primary_hits(16, 16)
    assert (prim == -1).all()


def test_sample_and_row_splits_compose(rt):
    """spp-split and tile-split: per-sample values do not depend on how the sample
    range or the rows are cut; ACCUMULATE adds passes in order."""
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    W, H = 48, 40
    full, rays_full = rt.render(W, H, 6, 8, seed=9)
    acc = rt.alloc(W * H * 16).zero()
    rays = rt.alloc(8).zero()
    rt.render_into(W, H, acc, 2, 8, seed=9, spp_begin=0, rays=rays)
    rt.render_into(W, H, acc, 4, 8, seed=9, spp_begin=2, rays=rays, flags=1)
    got = acc.to_numpy(np.float32, (H, W, 4))
    assert int(rays.to_numpy(np.uint64, (1,))[0]) == rays_full
    assert (got[..., 3] == 6).all()
    np.testing.assert_allclose(got[..., :3], full[..., :3], rtol=2e-6, atol=1e-6)   # (a+b)+(c+..) vs sequential
    acc.zero()
    rt.render_into(W, H, acc, 6, 8, seed=9, y_begin=0, y_count=23)
    rt.render_into(W, H, acc, 6, 8, seed=9, y_begin=23, y_count=17)
    assert O.same_bits_or_nan(acc.to_numpy(np.float32, (H, W, 4)), full)
    # interleaved 8-row bands (rt_render_bands): the shares of 3 "ranks" are disjoint and tile the frame;
    # a ragged height (last band of 3 rows), more ranks than bands, and the black (depth 0) path
    for Hb, world in [(H, 3), (43, 2), (43, 8)]:
        fullb, rays_b = rt.render(W, Hb, 5, 8, seed=9)
        total = np.zeros_like(fullb)
        n = 0
        for r in range(world):
            part, nr = rt.render(W, Hb, 5, 8, seed=9, band_first=r, band_stride=world)
            rows = np.zeros(Hb, bool)
            for b in range(r, (Hb + 7) // 8, world):
                rows[8 * b:8 * b + 8] = True
            assert (part[~rows] == 0).all() and (part[rows][..., 3] == 5).all()
            total += part
            n += nr
        assert n == rays_b and O.same_bits_or_nan(total, fullb)
    part, nr = rt.render(W, H, 4, 0, seed=9, band_first=1, band_stride=2)
    assert nr == 0 and (part[8:16, :, 3] == 4).all() and (part[0:8] == 0).all() and (part[..., :3] == 0).all()


def test_render_pixels_equals_the_frame_at_those_pixels(rt):
    """rt_render_pixels (RayView::renderOneRay for a list of random pixels, rayview.cpp:67-87): the listed
    pixels get exactly the sums rt_render gives them, nothing else is touched; duplicates, a ragged count
    (not a multiple of 64), accumulate, the black path and an empty list."""
    sc = scenes.get_scene("random", half=4)
    rt.upload_scene(sc)
    W, H, spp, depth = 96, 54, 3, 8
    full, _ = rt.render(W, H, spp, depth, seed=21)
    ofull, _ = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=21, math_mode=1)
    assert O.same_bits_or_nan(full, ofull)
    rng = np.random.default_rng(5)
    px = rng.integers(0, W * H, 1000 + 37).astype(np.int32)     # with duplicates
    acc = rt.alloc(W * H * 16).zero()
    rays = rt.alloc(8).zero()
    rt.render_pixels(W, H, acc, px, spp, depth, seed=21, rays=rays)
    got = acc.to_numpy(np.float32, (H, W, 4)).reshape(-1, 4)
    mask = np.zeros(W * H, bool); mask[px] = True
    assert O.same_bits_or_nan(got[mask], full.reshape(-1, 4)[mask])
    assert (got[~mask] == 0).all()
    assert int(rays.to_numpy(np.uint64, (1,))[0]) > 0
    # a second pass over other samples, accumulated: equals the 2-pass frame at those pixels
    rt.render_pixels(W, H, acc, np.unique(px), 2, depth, seed=21, flags=1, spp_begin=spp)
    two = rt.alloc(W * H * 16).zero()
    rt.render_into(W, H, two, spp, depth, seed=21)
    rt.render_into(W, H, two, 2, depth, seed=21, spp_begin=spp, flags=1)
    got = acc.to_numpy(np.float32, (H, W, 4)).reshape(-1, 4)
    assert O.same_bits_or_nan(got[mask], two.to_numpy(np.float32, (H, W, 4)).reshape(-1, 4)[mask])
    # depth 0: black samples, no rays; empty list: nothing happens
    acc.zero(); rays.zero()
    rt.render_pixels(W, H, acc, px[:10], 4, 0, seed=21, rays=rays)
    got = acc.to_numpy(np.float32, (H, W, 4)).reshape(-1, 4)
    assert int(rays.to_numpy(np.uint64, (1,))[0]) == 0 and (got[px[:10], 3] == 4).all() and (got[:, :3] == 0).all()
    rt.render_pixels(W, H, acc, np.zeros(0, np.int32), 4, 8, seed=21)


def test_tonemap_matches_oracle(rt):
    import ctypes as C
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    W, H = 80, 45
    acc, _ = rt.render(W, H, 4, 8, seed=2)
    acc[3, 4, 0] = np.nan; acc[5, 6, 1] = 1e9; acc[7, 8, 2] = 0.0
    img = rt.tonemap(W, H, acc)
    ref = np.empty((H, W), np.uint32)
    O.lib().orc_tonemap_rgb32(W, H, O.P(np.ascontiguousarray(acc), C.c_float), 0, ref.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert np.array_equal(img, ref)
    g = O.golden("materials")   # reference rgbPerSamples values, one row image
    tin, ns, tout = g["tone_in"], g["tone_n"], g["tone_out"]
    a4 = np.zeros((1, tin.shape[0], 4), np.float32); a4[0, :, :3] = tin; a4[0, :, 3] = ns
    img = rt.tonemap(tin.shape[0], 1, a4)[0, ::-1]
    want = (0xff000000 | (tout[:, 0].astype(np.uint32) << 16) | (tout[:, 1].astype(np.uint32) << 8) | tout[:, 2].astype(np.uint32))
    assert np.array_equal(img, want.astype(np.uint32))


def test_device_arithmetic_is_ieee(rt):
    """The bit-exactness argument rests on correctly rounded FP32/FP64 +,-,*,/,sqrt on gfx950 with
    contraction off. Probe it end to end: random one-sphere hits at many magnitudes must give the
    same double t as the CPU."""
    rng = np.random.default_rng(3)
    n = 50000
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    sc = scenes.Scene(mats, [scenes.Shape("sphere", (0.3, -0.2, 0.1, 1.7), 0),
                             scenes.Shape("xz_rect", (-3.3, 2.9, -1.7, 4.1, -2.45), 0)], None, scenes.rayview_camera())
    rt.upload_scene(sc, set_camera=False)
    scale = np.exp2(rng.integers(-12, 12, (n, 1))).astype(np.float32)
    ro = (rng.standard_normal((n, 3)).astype(np.float32) * np.float32(3.0))
    rd = (-ro + rng.standard_normal((n, 3)).astype(np.float32)) * scale
    got = rt.hit_rays(ro, rd, 1e-9, np.inf)
    ref = O.hit_rays(sc, ro, rd, 1e-9, np.inf)
    assert np.array_equal(got[0], ref[0]) and (got[0] >= 0).mean() > 0.5
    assert O.bits_equal(got[1], ref[1])
    assert O.same_bits_or_nan(got[2], ref[2]) and O.same_bits_or_nan(got[3], ref[3])


def test_cpp_dropin_host_renderAll(rt, tmp_path):
    """The C++ mirror of the reference headers (ray_amd/host) builds normal_scene with the reference's
    own class names and renders through renderAll(); its accumulators equal the oracle's."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ray_amd", "host", "render_demo")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = tmp_path / "accum.bin"
    r = subprocess.run([exe, "96", "54", "4", "8", "77", str(out)], capture_output=True, text=True, cwd=tmp_path, timeout=100)
    assert r.returncode == 0, r.stderr
    acc = np.fromfile(out, np.float32).reshape(54, 96, 4)
    sc = scenes.builtin_scene()
    oacc, _ = O.render(sc, 96, 54, 4, 8, rng_mode=1, seed64=77, math_mode=1)
    assert O.same_bits_or_nan(acc, oacc)
    assert "centre ray hits prim 0" in r.stdout


def test_progressive_slices_continue_the_sample_sequence(rt):
    """RayView::go on the GPU: time slices continue full-frame passes in sample order
    (RT_RENDER_CONTINUE); after any number of slices the accumulators hold the in-order sum of samples [0, n) of every
    pixel -- the bits of ONE render of n samples, whatever the passes were -- and reset() (scene mutation, as the
    sphere-Y slider does) starts over."""
    from ray_amd import api
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    W, H, depth = 96, 54, 8
    pv = api.Progressive(rt, W, H, depth, seed=21)
    n1 = pv.go(budget_ms=0.5)
    n2 = pv.go(budget_ms=2.0)
    assert 1 <= n1 < n2
    acc = pv.accumulators()
    assert (acc[..., 3] == n2).all()
    oacc, _ = O.render(sc, W, H, n2, depth, rng_mode=1, seed64=21, math_mode=1, n_threads=8)
    assert O.same_bits_or_nan(acc, oacc)          # passes continue the sum in sample order: no tolerance
    img = pv.image()
    assert img.shape == (H, W) and (img >> 24 == 0xff).all()
    moved = scenes.builtin_scene(sphere_y=0.8)
    pv.reset(moved)
    n3 = pv.go(budget_ms=0.2)
    acc3 = pv.accumulators()
    o3, _ = O.render(moved, W, H, n3, depth, rng_mode=1, seed64=21, math_mode=1, n_threads=8)
    assert (acc3[..., 3] == n3).all()
    assert O.same_bits_or_nan(acc3, o3)


@pytest.mark.parametrize("name,kw,W,H,rows", [
    ("random", {"half": 11}, 3840, 2160, (5, 1080, 2154)),          # BASELINE.json configs[2]
    ("tris", {"n": 71}, 1920, 1080, (400, 900)),                    # configs[3]
    ("mixed", {"half": 11, "n": 71}, 3840, 2160, (1500,)),          # configs[4]
    ("builtin", {}, 1920, 1080, (3, 540, 1079)),                    # configs[1]
])
def test_full_size_frame_properties(rt, name, kw, W, H, rows):
    """BASELINE.json's GPU configs at their full frame sizes (the oracle needs hours there): every scan of a
    1-spp frame equals the list-order loop (RT_RENDER_VERIFY), a frame cut into two sample ranges or
    into two row bands reproduces the uncut frame, and the sample count lands in .w."""
    sc = scenes.get_scene(name, **kw)
    rt.upload_scene(sc)
    depth = 8
    full, rays = rt.render(W, H, 2, depth, seed=0xC0FFEE)
    assert (full[..., 3] == 2).all() and np.isfinite(full).all() and rays > 2 * W * H
    _, rays_v = rt.render(W, H, 1, depth, seed=0xC0FFEE, flags=4)
    assert rt.verify_mismatches() == 0 and rays_v > W * H
    acc = rt.alloc(W * H * 16).zero()
    rt.render_into(W, H, acc, 1, depth, seed=0xC0FFEE, spp_begin=0)
    rt.render_into(W, H, acc, 1, depth, seed=0xC0FFEE, spp_begin=1, flags=1)
    assert O.same_bits_or_nan(acc.to_numpy(np.float32, (H, W, 4)), full)        # 2 samples: (a) + (b) is the same sum
    acc.zero()
    rt.render_into(W, H, acc, 2, depth, seed=0xC0FFEE, y_begin=0, y_count=H // 2 + 1)
    rt.render_into(W, H, acc, 2, depth, seed=0xC0FFEE, y_begin=H // 2 + 1, y_count=H - (H // 2 + 1))
    assert O.same_bits_or_nan(acc.to_numpy(np.float32, (H, W, 4)), full)
    acc.free()
    # a few rows of the frame against the oracle (bit-exact radiance sums and ray counts at full-size pixel coordinates)
    for y in rows:
        orow = np.zeros((H, W, 4), np.float32)
        O.render(sc, W, H, 2, depth, rng_mode=1, seed64=0xC0FFEE, math_mode=1, y_begin=y, y_count=1, accum=orow)
        assert O.same_bits_or_nan(full[y], orow[y])


def test_gpu_image_converges_to_the_reference_image(rt):
    """The only comparison that does not go through the oracle: the reference's OWN 4096-spp image
    (its minstd_rand0 streams, its libm) against 16384 spp from the HIP path (counter RNG, portable
    sin/cos). Same estimator, independent random numbers: frame mean within 0.5 %, per-pixel RMS
    difference at the level of the reference image's own 4096-spp noise."""
    g = O.golden("mean_builtin_160x90")
    ref = g["sum"] / 4096.0
    sc = golden_scenes()["builtin"]
    rt.upload_scene(sc)
    acc, _ = rt.render(160, 90, 16384, 8, seed=0x5EED)
    img = acc[..., :3] / 16384.0
    assert abs(img.mean() - ref.mean()) / ref.mean() < 0.005
    rms = float(np.sqrt(np.mean((img - ref) ** 2)))
    assert rms < 0.02, rms
    # per-channel means too (a colour cast would hide in the overall mean)
    for c in range(3):
        assert abs(img[..., c].mean() - ref[..., c].mean()) / ref[..., c].mean() < 0.01


def test_grazing_and_far_rays_match_oracle(rt):
    """Stress of the conservative sphere filter: rays built to pass within a few ulp of a
    sphere's silhouette, from origins 1 to 1e5 units away, with directions scaled over 12 binades, and
    rays that start on a sphere's surface. hitFromList through the filter + exact path must still equal
    the oracle's list-order loop bit for bit (primitive, t, hit record)."""
    sc = scenes.get_scene("random", half=11)
    rt.upload_scene(sc, set_camera=False)
    flat = sc.flatten()
    sph = flat["prim_data"][flat["prim_kind"] == 0][:, :4]
    rng = np.random.default_rng(11)
    n = 60000
    pick = rng.integers(0, sph.shape[0], n)
    c, r = sph[pick, :3], sph[pick, 3:4]
    # a point on the silhouette circle seen from the origin side: p = c + r * u with u a random unit vector
    u = rng.standard_normal((n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    # tangent direction at p: any vector orthogonal to u
    w = rng.standard_normal((n, 3)); w -= (w * u).sum(1, keepdims=True) * u; w /= np.linalg.norm(w, axis=1, keepdims=True)
    eps = rng.choice([0.0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-5, -1e-5, 1e-3, -1e-3], n)[:, None]   # inside / outside the rim
    p = c + r * (1.0 + eps) * u
    dist = np.exp(rng.uniform(0.0, np.log(1e5), n))[:, None]
    ro = (p - w * dist).astype(np.float32)
    scale = np.exp2(rng.integers(-6, 7, (n, 1))).astype(np.float64)
    rd = (w * scale).astype(np.float32)
    # a quarter of the rays start ON a sphere (bounce rays do): origin = surface point, random direction
    k = n // 4
    ro[:k] = (c[:k] + r[:k] * u[:k]).astype(np.float32)
    rd[:k] = rng.standard_normal((k, 3)).astype(np.float32)
    got = rt.hit_rays(ro, rd, 0.001, np.inf)
    ref = O.hit_rays(sc, ro, rd, 0.001, np.inf)
    assert np.array_equal(got[0], ref[0])
    assert O.bits_equal(got[1], ref[1])
    for a, b in zip(got[2:], ref[2:]):
        assert O.same_bits_or_nan(a, b)
    assert 0.05 < (got[0] >= 0).mean() < 0.999


def test_batch_boxes_edge_cases(rt):
    """Stress of the batch / word boxes: clustered spheres in many batches, exact
    duplicates at distant list positions (ties go to the later index, possibly in another batch),
    rays parallel to one or two axes (direction components exactly +-0), rays that run along box
    faces, rays starting inside a cluster, directions scaled over 30 binades, negative t_min and a finite
    t_max. hitFromList through boxes + filter + exact level equals the oracle's list-order loop."""
    rng = np.random.default_rng(23)
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    shapes = [scenes.Shape("sphere", (0.0, -1000.0, 0.0, 1000.0), 0)]
    centres = []
    for cl in range(12):                                   # 12 clusters x 40 spheres, flat and tall ones
        base = rng.uniform(-20, 20, 3) * np.array([1.0, 0.2, 1.0]) + np.array([0.0, 4.0, 0.0])
        ext = rng.choice([0.5, 2.0, 6.0], 3)
        for _ in range(40):
            c = np.float32(base + rng.uniform(-1, 1, 3) * ext)
            centres.append((float(c[0]), float(c[1]), float(c[2]), float(np.float32(rng.uniform(0.05, 0.6)))))
    grid = [(float(x), 0.25, float(z), 0.25) for x in range(-8, 8) for z in range(-8, 8)]   # touching neighbours on a lattice
    allsph = centres + grid
    allsph += [allsph[i] for i in rng.integers(0, len(allsph), 60)]                          # exact duplicates, later in the list
    for s in allsph:
        shapes.append(scenes.Shape("sphere", s, 0))
    sc = scenes.Scene(mats, shapes, None, scenes.rayview_camera())
    rt.upload_scene(sc, set_camera=False)
    n = 40000
    sph = np.array(allsph, np.float64)
    pick = rng.integers(0, sph.shape[0], n)
    target = sph[pick, :3] + rng.uniform(-1, 1, (n, 3)) * sph[pick, 3:4] * 1.2
    rd = rng.standard_normal((n, 3))
    # axis-parallel rays: zero one or two components (with both signs of zero)
    z1 = rng.integers(0, 3, n); z2 = rng.integers(0, 3, n)
    kind = rng.integers(0, 5, n)
    for k in range(3):
        rd[(kind == 1) & (z1 == k), k] = 0.0
        rd[(kind == 2) & ((z1 == k) | (z2 == k)), k] = -0.0
    rd[(rd == 0).all(1)] = (1.0, 0.0, 0.0)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    dist = np.exp(rng.uniform(np.log(0.01), np.log(200.0), n))[:, None]
    ro = target - rd * dist
    # kind 3: lattice rays running exactly along the faces of the grid spheres' boxes (y = 0 / 0.5, x = k +- 0.25)
    m3 = kind == 3
    ro[m3, 1] = rng.choice([0.0, 0.5, 0.25], m3.sum())
    ro[m3, 0] = rng.integers(-8, 8, m3.sum()) + rng.choice([-0.25, 0.25, 0.0], m3.sum())
    ro[m3, 2] = -30.0
    rd[m3] = (0.0, 0.0, 1.0)
    # kind 4: origin on a sphere's surface, random direction (bounce rays)
    m4 = kind == 4
    u = rng.standard_normal((n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    ro[m4] = (sph[pick, :3] + sph[pick, 3:4] * u)[m4]
    scale = np.exp2(rng.integers(-15, 16, (n, 1))).astype(np.float64)
    ro = ro.astype(np.float32); rd = (rd * scale).astype(np.float32)
    hits = 0
    for t_min, t_max in [(0.001, np.inf), (-5.0, np.inf), (0.0, 3.0), (0.25, 1.0e4)]:
        got = rt.hit_rays(ro, rd, t_min, t_max)
        ref = O.hit_rays(sc, ro, rd, t_min, t_max)
        assert np.array_equal(got[0], ref[0]), (t_min, t_max, np.flatnonzero(got[0] != ref[0])[:10])
        assert O.bits_equal(got[1], ref[1])
        for a, b in zip(got[2:], ref[2:]):
            assert O.same_bits_or_nan(a, b)
        hits += int((got[0] > 0).sum())
    assert hits > n                                         # plenty of hits on the small spheres, not only the ground
    # duplicates: some winning index must be one of the appended copies (ties go to the later entry)
    first_dup = 1 + len(centres) + len(grid)
    assert (rt.hit_rays(ro, rd, 0.001, np.inf)[0] > first_dup).any()


@pytest.mark.parametrize("n_sph", [15, 16, 17, 31, 33, 63, 65, 129, 513, 1025, 3000])
def test_group_layout_sizes(rt, n_sph):
    """Sphere counts around every padding boundary of the scan's layout (16-item batches, 4-batch words,
    16-word chunks, the 16-item minimum of the filter group), mixed radii with a few huge / far spheres that
    go to the exact group: hitFromList equals the oracle's loop for random and for targeted rays."""
    rng = np.random.default_rng(1000 + n_sph)
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    c = rng.uniform(-10, 10, (n_sph, 3))
    r = np.exp(rng.uniform(np.log(0.02), np.log(0.8), n_sph))
    big = rng.random(n_sph) < 0.03
    r[big] *= 200.0
    c[big] *= 30.0
    shapes = [scenes.Shape("sphere", (float(np.float32(c[i, 0])), float(np.float32(c[i, 1])), float(np.float32(c[i, 2])), float(np.float32(r[i]))), 0)
              for i in range(n_sph)]
    sc = scenes.Scene(mats, shapes, None, scenes.rayview_camera())
    rt.upload_scene(sc, set_camera=False)
    n = 6000
    pick = rng.integers(0, n_sph, n)
    target = c[pick] + rng.uniform(-1, 1, (n, 3)) * r[pick, None]
    ro = rng.uniform(-25, 25, (n, 3))
    rd = target - ro
    rd[: n // 3] = rng.standard_normal((n // 3, 3))
    ro, rd = ro.astype(np.float32), rd.astype(np.float32)
    got = rt.hit_rays(ro, rd, 0.001, np.inf)
    ref = O.hit_rays(sc, ro, rd, 0.001, np.inf)
    assert np.array_equal(got[0], ref[0])
    assert O.bits_equal(got[1], ref[1])
    assert (got[0] >= 0).mean() > 0.3


@pytest.mark.parametrize("n_tri", [1, 31, 33, 257, 1023, 1100, 2500])
def test_triangle_group_layout_sizes(rt, n_tri):
    """Triangle soups (single-triangle items and quad halves that share an item, sizes around the batch /
    word / chunk boundaries of the triangle group, a few huge triangles for the exact group): hitFromList
    through boxes + item flags + Moeller-Trumbore equals the oracle's loop."""
    rng = np.random.default_rng(2000 + n_tri)
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    shapes = []
    cents = []
    while len(shapes) < n_tri:
        p = rng.uniform(-10, 10, 3)
        e1, e2 = rng.standard_normal(3) * 0.5, rng.standard_normal(3) * 0.5
        if rng.random() < 0.02:
            e1 *= 100.0; e2 *= 100.0
        v = [np.float32(p), np.float32(p + e1), np.float32(p + e2), np.float32(p + e1 + e2)]
        shapes.append(scenes.Shape("triangle", tuple(float(t) for t in (*v[0], *v[1], *v[2])), 0))
        cents.append((v[0] + v[1] + v[2]) / 3)
        if rng.random() < 0.5 and len(shapes) < n_tri:     # the other half of the quad, consecutive in the list
            shapes.append(scenes.Shape("triangle", tuple(float(t) for t in (*v[2], *v[1], *v[3])), 0))
            cents.append((v[2] + v[1] + v[3]) / 3)
    sc = scenes.Scene(mats, shapes, None, scenes.rayview_camera())
    rt.upload_scene(sc, set_camera=False)
    n = 6000
    cents = np.array(cents, np.float64)
    pick = rng.integers(0, n_tri, n)
    ro = rng.uniform(-20, 20, (n, 3))
    rd = cents[pick] + rng.uniform(-0.2, 0.2, (n, 3)) - ro
    rd[: n // 4] = rng.standard_normal((n // 4, 3))
    ro, rd = ro.astype(np.float32), rd.astype(np.float32)
    for t_min, t_max in [(0.001, np.inf), (-3.0, 40.0)]:
        got = rt.hit_rays(ro, rd, t_min, t_max)
        ref = O.hit_rays(sc, ro, rd, t_min, t_max)
        assert np.array_equal(got[0], ref[0])
        assert O.bits_equal(got[1], ref[1])
    assert (got[0] >= 0).mean() > 0.3


@pytest.mark.parametrize("n_rect", [15, 16, 17, 33, 160, 513, 1030])
def test_rect_flag_pass_sizes_and_edge_rays(rt, n_rect):
    """Rect soups around the boundaries of the rect flag pass (16-rect batches, 32-rect words, 512-rect chunks;
    below 16 the direct loop runs), all three orientations, some rects duplicated later in the list (ties go to
    the later entry). Rays: random, aimed at rect corners and edges (points on the boundary itself), starting ON
    a rect, axis-parallel with +-0 components (the reference's inf / NaN plane distances: list-order loop),
    negative t_min and finite t_max."""
    rng = np.random.default_rng(3000 + n_rect)
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    kinds = ["xy_rect", "xz_rect", "yz_rect"]
    shapes, geo = [], []
    while len(shapes) < n_rect:
        if geo and rng.random() < 0.1:
            kind, q = geo[rng.integers(0, len(geo))]                      # an exact duplicate
        else:
            kind = int(rng.integers(0, 3))
            a0, b0 = rng.uniform(-10, 10, 2)
            q = (float(np.float32(a0)), float(np.float32(a0 + rng.uniform(0.1, 4.0))), float(np.float32(b0)),
                 float(np.float32(b0 + rng.uniform(0.1, 4.0))), float(np.float32(rng.uniform(-10, 10))))
        geo.append((kind, q))
        shapes.append(scenes.Shape(kinds[kind], q, 0))
    sc = scenes.Scene(mats, shapes, None, scenes.rayview_camera())
    rt.upload_scene(sc, set_camera=False)

    def point_on(kind, q, fa, fb):
        a, b = q[0] + fa * (q[1] - q[0]), q[2] + fb * (q[3] - q[2])
        return [(a, b, q[4]), (a, q[4], b), (q[4], a, b)][kind]
    n = 8000
    pick = rng.integers(0, n_rect, n)
    frac = rng.choice([0.0, 1.0, 0.5, -1e-7, 1.0 + 1e-7, 0.3], (n, 2))    # corners, edges, just outside, inside
    target = np.array([point_on(*geo[pick[i]], frac[i, 0], frac[i, 1]) for i in range(n)])
    ro = rng.uniform(-14, 14, (n, 3))
    rd = target - ro
    k4 = n // 4
    rd[:k4] = rng.standard_normal((k4, 3))                                # random directions
    ro[k4:k4 + 500] = target[k4:k4 + 500]                                 # origin ON a rect
    rd[k4:k4 + 500] = rng.standard_normal((500, 3))
    ax = rng.integers(0, 3, 600)                                          # axis-parallel rays, both signs of zero
    rd[-600:] = rng.standard_normal((600, 3))
    rd[np.arange(n - 600, n), ax] = rng.choice([0.0, -0.0], 600)
    rd[-200:, (ax[-200:] + 1) % 3] = 0.0
    rd[(rd == 0).all(1)] = (0.0, 1.0, 0.0)
    ro, rd = ro.astype(np.float32), rd.astype(np.float32)
    for t_min, t_max in [(0.001, np.inf), (-2.0, 9.0)]:
        got = rt.hit_rays(ro, rd, t_min, t_max)
        ref = O.hit_rays(sc, ro, rd, t_min, t_max)
        assert np.array_equal(got[0], ref[0]), np.flatnonzero(got[0] != ref[0])[:10]
        for a, b in zip(got[1:], ref[1:]):
            assert O.same_bits_or_nan(a, b)
    assert (got[0] >= 0).mean() > 0.3


def test_triangle_known_answers_on_the_gpu(rt):
    """tests/tri_kat.py through rt_hit_rays: hand-computed unit-triangle hits, float64 geometry on random triangles
    (incl. near-in-plane rays from far away), and two triangles against the reference's `rectangle`."""
    import tri_kat

    def hit_rays(sc, ro, rd, t_min, t_max):
        rt.upload_scene(sc, set_camera=False)
        return rt.hit_rays(ro, rd, t_min, t_max)
    tri_kat.check_unit_triangle(hit_rays)
    for seed in (5, 6, 7):
        tri_kat.check_general_triangles(hit_rays, seed=seed)
    tri_kat.check_triangle_pair_equals_rectangle(hit_rays)
    tri_kat.check_t_min_that_is_no_float(hit_rays)


def test_render_pixels_duplicates_accumulate_and_stray_indices(rt):
    """rt_render_pixels with RT_RENDER_ACCUMULATE and a pixel listed k times (RayView::go draws its pixels at random,
    rayview.cpp:69-70, so repeats are the normal case): the pixel receives k times its sums and k times the sample
    count, deterministically; indices outside the frame are skipped, not written."""
    from ray_amd import api
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    W, H, spp, depth = 64, 36, 3, 8
    full, _ = rt.render(W, H, spp, depth, seed=4)
    base = np.arange(0, W * H, 7, dtype=np.int32)
    reps = (np.arange(base.size) % 4) + 1                                   # 1..4 listings per pixel
    px = np.repeat(base, reps)
    np.random.default_rng(0).shuffle(px)
    stray = np.array([-1, W * H, W * H + 5, 2 ** 31 - 1, -2 ** 31], np.int32)
    px = np.concatenate([stray[:2], px, stray[2:]])
    for trial in range(3):                                                  # same result every time
        acc = rt.alloc((W * H + 16) * 16).zero()                           # 16 guard entries behind the frame
        rt.render_pixels(W, H, acc, px, spp, depth, seed=4, flags=api.RENDER_ACCUMULATE)
        got = acc.to_numpy(np.float32, (W * H + 16, 4))
        assert (got[W * H:] == 0).all()
        want = np.zeros((W * H, 4), np.float32)
        f = full.reshape(-1, 4)
        for k in range(1, 5):
            sel = base[reps == k]
            v = np.zeros((sel.size, 4), np.float32)
            for _ in range(k):
                v = v + f[sel]
            want[sel] = v
        assert O.same_bits_or_nan(got[:W * H], want)
        # overwrite mode with the same list: every listed pixel holds the frame's value, strays ignored
        acc.zero()
        rt.render_pixels(W, H, acc, px, spp, depth, seed=4)
        got = acc.to_numpy(np.float32, (W * H + 16, 4))
        assert O.same_bits_or_nan(got[base], f[base]) and (got[W * H:] == 0).all()
        acc.zero()
        rt.render_pixels(W, H, acc, px, 2, 0, seed=4, flags=api.RENDER_ACCUMULATE)     # black path
        got = acc.to_numpy(np.float32, (W * H + 16, 4))
        assert (got[base, 3] == 2 * reps).all() and (got[:, :3] == 0).all() and (got[W * H:] == 0).all()
        acc.free()
    assert rt.lib.rt_tonemap(70000, 70000, 1, 1, 1) != 0 and b"bad size" in rt.lib.rt_last_error()


def test_nowait_launch_and_contexts(rt):
    """RT_RENDER_NOWAIT returns with the kernel queued; rt_sync (or the next call) waits; results equal the blocking
    call's. rt_set_device only accepts devices that rt_init has given a context."""
    from ray_amd import api
    sc = scenes.get_scene("random", half=4)
    rt.upload_scene(sc)
    W, H, spp, depth = 320, 180, 4, 8
    want, want_rays = rt.render(W, H, spp, depth, seed=3)
    acc, rays = rt.alloc(W * H * 16).zero(), rt.alloc(8).zero()
    rt.render_into(W, H, acc, 2, depth, seed=3, rays=rays, flags=api.RENDER_NOWAIT)
    rt.render_into(W, H, acc, 2, depth, seed=3, rays=rays, spp_begin=2, flags=api.RENDER_NOWAIT | api.RENDER_ACCUMULATE)
    rt.sync()
    ms, n = rt.last_kernel_ms()
    assert ms > 0 and n == 1
    assert int(rays.to_numpy(np.uint64, (1,))[0]) == want_rays
    got = acc.to_numpy(np.float32, (H, W, 4))
    assert (got[..., 3] == 4).all()
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=1e-6)
    acc.zero()
    rt.render_into(W, H, acc, spp, depth, seed=3, flags=api.RENDER_NOWAIT)
    assert O.same_bits_or_nan(acc.to_numpy(np.float32, (H, W, 4)), want)        # the copy waits for the launch
    acc.free(); rays.free()
    assert rt.lib.rt_set_device(0) == 0
    assert rt.lib.rt_set_device(63) != 0 and b"no context" in rt.lib.rt_last_error()
    assert rt.lib.rt_set_device(-1) != 0


def test_kernels_are_ordered_with_torch_work_on_a_handed_in_stream(rt):
    """rt_set_stream (the stream-ordering rule of include/ray_amd.h): with torch's current stream handed to the
    library, a tensor.zero_() queued before a render and a tensor op queued after it are ordered with the kernel.
    A large low-spp frame rendered into a torch tensor that is re-filled with garbage and zeroed asynchronously
    before every pass: bands of other 'ranks' stay zero, the own bands hold the frame."""
    import torch
    sc = scenes.get_scene("random", half=11)
    rt.upload_scene(sc)
    W, H, depth = 3840, 2160, 8
    want, _ = rt.render(W, H, 1, depth, seed=0xC0FFEE)
    dev = torch.device("cuda", 0)
    acc = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)
    try:
        for stream in (torch.cuda.current_stream(dev), side):
            rt.set_stream(stream)
            with torch.cuda.stream(stream):
                for first in (0, 1, 0):
                    acc.fill_(float("nan"))
                    big = torch.randn(64 * 1024 * 1024, device=dev)       # some queued work in front
                    big.mul_(2.0)
                    acc.zero_()
                    rt.render_into(W, H, acc, 1, depth, seed=0xC0FFEE, band_first=first, band_stride=2)
                    doubled = acc * 2.0                                    # queued behind the kernel on the same stream
                    got = acc.cpu().numpy()
                    mine = np.zeros(H, bool)
                    for b in range(first, (H + 7) // 8, 2):
                        mine[8 * b:8 * b + 8] = True
                    assert (got[~mine] == 0).all()
                    assert O.same_bits_or_nan(got[mine], want[mine])
                    assert O.same_bits_or_nan(doubled.cpu().numpy()[mine], 2.0 * want[mine])
                    del big, doubled
    finally:
        rt.set_stream(None)
    # back on the library's own stream: the documented alternative is to synchronise first
    acc.fill_(float("nan")); acc.zero_()
    torch.cuda.synchronize(dev)
    rt.render_into(W, H, acc, 1, depth, seed=0xC0FFEE, band_first=1, band_stride=2)
    assert O.same_bits_or_nan(acc.cpu().numpy()[8:16], want[8:16]) and (acc.cpu().numpy()[0:8] == 0).all()


def test_two_rank_job_drives_the_hip_library():
    """tests/dist_worker.py --device under torch.distributed.run with two ranks on this box's one GPU (gloo moves the
    tensors; RCCL refuses two ranks on one device): spp-split + all-reduce, tile-split + band all-gather, rows-split,
    all rendered by the HIP library into torch tensors, against the oracle's whole frame."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(root, "tests", "dist_worker.py"), "--device"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_reference_signature_caller_renderOneRay(rt, tmp_path):
    """ray_amd/host/rayview_tick.cpp is RayView::renderOneRay (rayview.cpp:67-87) written against the mirror headers
    with the reference's own calls -- cam.get_ray(u, v), six-argument calc::ray_color, calc::rgbPerSamples, no session
    anywhere. Its log is checked sample by sample: (u, v) from the reference's gen01 stream, rays equal to the oracle's
    camera::get_ray on the same engines, every colour equal to the oracle's ray_color on the stream the call drew from,
    pixel sums and tonemapped pixels as the reference computes them; the scene is uploaded once per change, not per call."""
    import ctypes as C
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ray_amd", "host", "rayview_tick")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    ticks, spp, depth, seeds = 40, 3, 10, (5, 6, 7)
    log = tmp_path / "tick.bin"
    r = subprocess.run([exe, str(ticks), str(spp), str(depth), *[str(s) for s in seeds], str(log)], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    # uploads: scene, moved sphere, the list alone for hitFromList (no lights), one shape alone for shape::hit
    m = re.search(r"uploads (\d+), centre ray: list prim (-?\d+) t=(\S+), first sphere alone t=(\S+), world box 1 \((.*?)\)-\((.*?)\) hit 1", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == 4 and int(m.group(2)) == 0 and float(m.group(3)) == 1.0 and float(m.group(4)) == 1.0   # the tangent hit of SURVEY.md 8c
    assert [float(x) for x in m.group(5).split()] == [-100.0, np.float32(-200.2), -101.0]      # surrounding_box of the five shapes
    rec = np.dtype([("uv", "<f4", 2), ("o", "<f4", 3), ("d", "<f4", 3), ("dl2", "<f4"), ("stream", "<u8"), ("c", "<f4", 3)])
    tick = np.dtype([("xy", "<i4", 2), ("s", rec, spp), ("sum", "<f4", 3), ("px", "<f4", 3)])
    assert rec.itemsize == 56
    t = np.fromfile(log, tick)
    assert t.shape == (2 * ticks,)
    s = t["s"].reshape(-1)
    n = s.shape[0]
    assert np.array_equal(s["stream"], np.arange(n, dtype=np.uint64))        # call k draws from stream k
    # (u, v): two gen01 draws per sample, (rand + xi) / 800 and / 450 in FP32 (rayview.cpp:80-81)
    xi = np.empty(2 * n, np.float32)
    O.lib().orc_refrng_fill(seeds[1], 0, 0.0, 1.0, 0, 2 * n, O.P(xi, C.c_float))
    xy = np.repeat(t["xy"], spp, axis=0)
    assert xy[:, 0].min() >= 0 and xy[:, 0].max() <= 800 and xy[:, 1].max() <= 450
    u = ((xy[:, 0].astype(np.float32) + xi[0::2]) / np.float32(800)).astype(np.float32)
    v = ((xy[:, 1].astype(np.float32) + xi[1::2]) / np.float32(450)).astype(np.float32)
    assert O.bits_equal(s["uv"][:, 0].copy(), u) and O.bits_equal(s["uv"][:, 1].copy(), v)
    # rays: camera::get_ray on the reference's gen11 stream (oracle restatement, pinned by camera.npz)
    sc = scenes.builtin_scene()
    cam = O.camera_of(sc)
    ro, rd, dl2 = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.float32)
    O.lib().orc_camera_rays(C.byref(cam), seeds[0], seeds[1], seeds[2], n, O.P(np.ascontiguousarray(s["uv"]), C.c_float), O.P(ro, C.c_float),
                            O.P(rd, C.c_float), O.P(dl2, C.c_float))
    assert O.bits_equal(np.ascontiguousarray(s["o"]), ro) and O.bits_equal(np.ascontiguousarray(s["d"]), rd)
    assert O.bits_equal(np.ascontiguousarray(s["dl2"]), dl2)
    # colours: the oracle's ray_color for ray k on counter stream (k, 0, 0); the second half sees the moved sphere
    moved = scenes.builtin_scene(sphere_y=0.8)
    for k in range(n):
        want = O.trace_rays(sc if k < n // 2 else moved, ro[k:k + 1], rd[k:k + 1], depth, seed=k)
        assert O.same_bits_or_nan(s["c"][k], want[0]), k
    # pixel sums in sample order, and calc::rgbPerSamples of them
    acc = np.zeros((2 * ticks, 3), np.float32)
    for j in range(spp):
        acc = acc + t["s"]["c"][:, j]
    assert O.same_bits_or_nan(t["sum"], acc)
    tone = np.empty((2 * ticks, 3), np.float32)
    O.lib().orc_tonemap(2 * ticks, O.P(np.ascontiguousarray(t["sum"]), C.c_float), O.P(np.full(2 * ticks, spp, np.int32), C.c_int32), O.P(tone, C.c_float))
    assert O.same_bits_or_nan(t["px"], tone)


def test_shape_hit_on_the_device_fills_u_v_like_the_reference(tmp_path):
    """shape::hit of the mirror headers (device closest-hit kernel, then hit_record::u, v on the host as sphere::get_sphere_uv and the
    rects' (x - x0) / (x1 - x0) compute them) on every 97th case of tests/golden/shapehits.npz -- the compiled reference's own
    sphere::hit / xy/xz/yz_rect::hit outputs: has, t, p, normal, front_face, u, v bit for bit."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ray_amd", "host", "mirror_kat")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    g = O.golden("shapehits")
    step = 97    # prime against the period of the case generator (every 100th rect case is a miss)
    for pre, kind, prm in (("sph", np.zeros(24000, np.int32), np.pad(g["sph_prm"].astype(np.float64), ((0, 0), (0, 1)))),
                           ("rect", (g["rect_axis"] + 1).astype(np.int32), g["rect_prm"])):
        sel = np.arange(0, kind.shape[0], step)
        n = sel.shape[0]
        with open(tmp_path / "in.bin", "wb") as f:
            for a in (kind, prm, g[pre + "_ray_o"], g[pre + "_ray_d"], g[pre + "_tmax"]):
                f.write(np.ascontiguousarray(a[sel]).tobytes())
        r = subprocess.run([exe, "shapehits_dev", str(n), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr
        b, pos = np.fromfile(tmp_path / "out.bin", np.uint8), 0

        def take(dtype, shape):
            nonlocal pos
            k = int(np.prod(shape)) * np.dtype(dtype).itemsize
            out = b[pos:pos + k].view(dtype).reshape(shape)
            pos += k
            return out
        has, t, pp, nrm, ff, uv = take(np.uint8, (n,)), take(np.float64, (n,)), take(np.float32, (n, 3)), take(np.float32, (n, 3)), take(np.uint8, (n,)), take(np.float64, (n, 2))
        assert np.array_equal(has, g[pre + "_has"][sel]) and 0.05 < has.mean() < 0.95, pre
        h = has.astype(bool)
        assert O.bits_equal(t[h], g[pre + "_t"][sel][h]) and O.bits_equal(pp[h], g[pre + "_p"][sel][h]) and O.bits_equal(nrm[h], g[pre + "_n"][sel][h]), pre
        assert np.array_equal(ff[h], g[pre + "_ff"][sel][h]) and O.bits_equal(uv[h], g[pre + "_uv"][sel][h]), pre


@pytest.mark.parametrize("split,shares", [("spp", 3), ("tile", 3), ("tile", 8), ("spp", 1)])
def test_cpp_multi_device_host(split, shares, tmp_path):
    """ray_amd/host/render_multi.cpp: one C++ process with one library context per device, shares queued with
    RT_RENDER_NOWAIT and collected in order -- the multi-GPU split below Python. On this one-GPU box the shares are
    dealt onto the one device; the merged frame must be the oracle's (bit-exact for disjoint bands, summation-order
    tolerance for the sample split)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ray_amd", "host", "render_multi")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    W, H, spp, depth, seed = 120, 67, 7, 8, 99
    out = tmp_path / "acc.bin"
    r = subprocess.run([exe, str(W), str(H), str(spp), str(depth), str(seed), str(shares), split, str(out)], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    acc = np.fromfile(out, np.float32).reshape(H, W, 4)
    oacc, orays = O.render(scenes.builtin_scene(), W, H, spp, depth, rng_mode=1, seed64=seed, math_mode=1, n_threads=8)
    assert f"{orays} rays" in r.stdout and (acc[..., 3] == spp).all()
    if split == "tile" or shares == 1:
        assert O.same_bits_or_nan(acc, oacc)
    else:
        np.testing.assert_allclose(acc[..., :3], oacc[..., :3], rtol=2e-6, atol=1e-6)


def test_4k_frame_has_the_same_bits_however_its_end_is_cut(rt):
    """render_impl's policy for the end of a frame (how many tiles go through the colour buffer -- six per resident wave on a frame
    of this size --, whole pixels or sample slices for the others) decides HOW the work is cut, never the result: the 485-sphere
    scene at 3840x2160, 32 samples, with no buffer, the default one and one smaller than the frame would like."""
    sc = scenes.get_scene("random", half=11)
    rt.upload_scene(sc)
    W, H, spp, depth = 3840, 2160, 32, 8
    acc = rt.alloc(W * H * 16)
    try:
        ref = None
        for mb, slices in ((0, (0, 16)), (2048, (8, 16)), (300, (8, 16)), (100, (3, 5))):
            rt.set_tail_limit(mb); rt.set_sample_slices(*slices)
            acc.zero()
            rt.render_into(W, H, acc, spp, depth, seed=3)
            got = acc.to_numpy(np.uint32, (H, W, 4)).copy()
            if ref is None:
                ref = got
                assert (got[..., 3].view(np.float32) == spp).all()
            assert np.array_equal(got, ref), (mb, slices)
    finally:
        rt.set_tail_limit(1400); rt.set_sample_slices(8, 16)
        acc.free()


def test_tiles_in_order_of_cost_leave_the_same_bits(rt):
    """rt_set_tile_order: the first sample as a pass of its own that counts every tile's rays, the tiles sorted by that
    count, the other samples continued with the heaviest tiles first. Mode 1 (the default) does it where the frame's end has little
    filler (this 1080p frame of 64 samples), mode 2 everywhere: accumulators, sample counts and ray counts are those of mode 0 --
    whole frames, interleaved bands, passes that continue one another, a frame without a colour buffer, a small frame."""
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    try:
        for W, H, spp, depth, kw in ((1920, 1080, 64, 10, {}), (1920, 1080, 64, 10, dict(band_first=1, band_stride=2)), (333, 187, 19, 6, {}),
                                    (1283, 721, 40, 8, dict(tail_mb=0)), (1283, 721, 33, 8, dict(passes=(5, 28))), (640, 360, 24, 8, dict(nowait=True))):
            kw = dict(kw)
            tail_mb, passes, nowait = kw.pop("tail_mb", 1400), kw.pop("passes", None), kw.pop("nowait", False)
            rt.set_tail_limit(tail_mb)
            acc = rt.alloc(W * H * 16)
            rays = rt.alloc(8)
            ref = None
            for mode in (0, 1, 2):
                rt.set_tile_order(mode)
                acc.zero(); rays.zero()
                if passes:
                    s0 = 0
                    for i, n in enumerate(passes):
                        rt.render_into(W, H, acc, n, depth, seed=9, rays=rays, spp_begin=s0, flags=api_flags_continue() if i else 0, **kw)
                        s0 += n
                elif nowait:                       # RT_RENDER_NOWAIT: the call returns with both launches queued; rt_sync waits
                    from ray_amd import api
                    rt.render_into(W, H, acc, spp, depth, seed=9, rays=rays, flags=api.RENDER_NOWAIT)
                    rt.sync()
                    assert rt.last_kernel_ms()[0] > 0
                else:
                    rt.render_into(W, H, acc, spp, depth, seed=9, rays=rays, **kw)
                got = (acc.to_numpy(np.uint32, (H, W, 4)).copy(), int(rays.to_numpy("uint64", (1,))[0]))
                if ref is None:
                    ref = got
                assert np.array_equal(got[0], ref[0]) and got[1] == ref[1], (W, H, spp, kw, mode)
            acc.free(); rays.free()
    finally:
        rt.set_tile_order(1); rt.set_tail_limit(1400)


def api_flags_continue():
    from ray_amd import api
    return api.RENDER_CONTINUE


def test_c_abi_tour_from_plain_cpp():
    """ray_amd/host/abi_tour.cpp: every entry point of include/ray_amd.h from plain C++ (no Python, no mirror headers) on a scene
    with every primitive and material kind -- rejected arguments are rejected and leave the context usable; rt_render_noalloc,
    rt_render_bands, two row ranges, two RT_RENDER_CONTINUE passes, every work-unit setting and the shading sort leave rt_render's
    bits; pixel lists short and long touch the listed pixels only; RT_RENDER_VERIFY counts no mismatch; the context survives
    rt_shutdown + rt_init. (tools/asan_run.sh runs the same program against a host-AddressSanitizer build of the library.)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ray_amd", "host", "abi_tour")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "abi tour ok" in r.stdout


def test_constant_medium_through_the_c_abi(rt):
    """constant_medium (shapes.h:374-455) in hitFromList for caller rays: the medium draws from the ray's own counter
    stream inside the scan, so which rays draw depends on list order; records carry the reference's 'arbitrary' normal
    (1,0,0) and front_face true. Against the oracle on the smoke scene and on media-only lists (sphere and box
    boundaries, origins inside, rays parallel to box faces, negative t_min, finite t_max). Plus the upload rules."""
    from ray_amd import capi
    rng = np.random.default_rng(17)
    iso = scenes.Material(scenes.MAT_ISOTROPIC, (0.9, 0.9, 0.9))
    lam = scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    only = scenes.Scene([iso, lam], [scenes.Shape("medium_sphere", (0.5, 0.25, -0.5, 1.5, 0.8), 0),
                                     scenes.Shape("sphere", (0.5, 0.25, -0.5, 0.5), 1),
                                     scenes.Shape("medium_box", (-2.0, -1.0, -3.0, 0.0, 1.5, 1.0, 2.0), 0),
                                     scenes.Shape("medium_box", (-1.0, -0.5, -1.0, 3.0, 0.5, 0.0, 0.05), 0)], None, scenes.rayview_camera())
    for sc in (scenes.smoke_scene(), only):
        rt.upload_scene(sc)
        n = 30000
        lo, hi = (-1.0, 7.0) if sc.name == "smoke" else (-4.0, 4.0)
        ro = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
        rd = (rng.uniform(lo, hi, (n, 3)) - ro).astype(np.float32) * rng.choice([0.1, 1.0, 6.0], (n, 1)).astype(np.float32)
        rd[::17, 0] = 0.0                      # parallel to two faces of every box
        rd[5::29] = 0.0                        # zero direction
        seen = 0
        for t_min, t_max in [(0.001, np.inf), (-3.0, np.inf), (0.2, 2.5)]:
            got = rt.hit_rays(ro, rd, t_min, t_max)
            ref = O.hit_rays(sc, ro, rd, t_min, t_max)
            assert np.array_equal(got[0], ref[0])
            assert O.bits_equal(got[1], ref[1])
            for a, b in zip(got[2:], ref[2:]):
                assert O.same_bits_or_nan(a, b)
            kinds = sc.flatten()["prim_kind"]
            med = (got[0] >= 0) & np.isin(kinds[np.maximum(got[0], 0)], (scenes.PRIM_MEDIUM_SPHERE, scenes.PRIM_MEDIUM_BOX))
            assert (got[3][med] == (1, 0, 0)).all() and (got[4][med] == 1).all()
            seen += int(med.sum())
        assert seen > n // 4
    # RT_RENDER_VERIFY has nothing to compare on such a scene; a medium must carry an isotropic material
    acc, rays = rt.alloc(16 * 16 * 16), rt.alloc(8)
    assert rt.lib.rt_render(16, 16, 0, 16, 0, 1, 4, 0, 4, acc.ptr, rays.ptr) != 0 and b"constant_medium" in rt.lib.rt_last_error()
    bad = scenes.Scene([lam, iso], [scenes.Shape("medium_sphere", (0, 0, 0, 1, 1.0), 1)], None, scenes.rayview_camera())
    flat = bad.flatten(); flat["prim_mat"][:] = 0
    assert rt.lib.rt_upload_scene(capi.SceneArrays(flat).ref()) != 0 and b"RT_MAT_ISOTROPIC" in rt.lib.rt_last_error()
    acc.free(); rays.free()
    # an isotropic SURFACE (a sphere whose material is isotropic) scatters uniformly and is read as specular
    surf = scenes.Scene([iso, lam], [scenes.Shape("sphere", (0.0, 0.0, -1.0, 0.5), 0), scenes.Shape("sphere", (0.0, -100.5, -1.0, 100.0), 1)],
                        scenes.LIGHT_TARGET, scenes.rayview_camera())
    rt.upload_scene(surf)
    a, r = rt.render(64, 36, 4, 8, seed=3)
    oa, orr = O.render(surf, 64, 36, 4, 8, rng_mode=1, seed64=3, math_mode=1)
    assert r == orr and O.same_bits_or_nan(a, oa)


def test_triangle_bvh_forced_on_small_and_odd_scenes(rt):
    """The per-lane triangle BVH (scan_triangles_bvh) is an alternative to the flat box levels that
    measures slower and is therefore off unless RT_TRI_BVH_MIN=<items> asks for it. RT_TRI_BVH_MIN=1 forces it onto tiny
    and odd-sized soups (1, 2, 3, 5 ... triangles: the one-leaf root, unbalanced halves), a huge value keeps it off: both
    paths must give the oracle's hitFromList, the analytic known answers included; and a frame of the 9800-triangle
    mesh through the BVH must equal the list-order loop ray for ray (RT_RENDER_VERIFY) and the flat path's frame."""
    import os
    import tri_kat
    rng = np.random.default_rng(77)
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    try:
        for bvh_min in ("1", "1000000"):
            os.environ["RT_TRI_BVH_MIN"] = bvh_min

            def hit_rays(sc, ro, rd, t_min, t_max):
                rt.upload_scene(sc, set_camera=False)
                return rt.hit_rays(ro, rd, t_min, t_max)
            tri_kat.check_unit_triangle(hit_rays)
            tri_kat.check_general_triangles(hit_rays, seed=8, n_tri=10)
            tri_kat.check_triangle_pair_equals_rectangle(hit_rays)
            for n_tri in (1, 2, 3, 5, 8, 63, 130):
                shapes, cents = [], []
                for _ in range(n_tri):
                    p = rng.uniform(-6, 6, 3)
                    e1, e2 = rng.standard_normal(3) * 0.8, rng.standard_normal(3) * 0.8
                    v = [np.float32(p), np.float32(p + e1), np.float32(p + e2)]
                    shapes.append(scenes.Shape("triangle", tuple(float(t) for t in (*v[0], *v[1], *v[2])), 0))
                    cents.append((v[0] + v[1] + v[2]) / 3)
                shapes.append(scenes.Shape("sphere", (0.0, -50.0, 0.0, 44.0), 0))        # something behind the soup
                sc = scenes.Scene(mats, shapes, None, scenes.rayview_camera())
                rt.upload_scene(sc, set_camera=False)
                n = 4000
                cents_a = np.array(cents, np.float64)
                ro = rng.uniform(-15, 15, (n, 3))
                rd = cents_a[rng.integers(0, n_tri, n)] + rng.uniform(-0.3, 0.3, (n, 3)) - ro
                rd[: n // 4] = rng.standard_normal((n // 4, 3))
                rd[::37, 1] = 0.0
                ro, rd = ro.astype(np.float32), rd.astype(np.float32)
                for t_min, t_max in [(0.001, np.inf), (-4.0, 30.0)]:
                    got = rt.hit_rays(ro, rd, t_min, t_max)
                    ref = O.hit_rays(sc, ro, rd, t_min, t_max)
                    assert np.array_equal(got[0], ref[0]), (bvh_min, n_tri)
                    assert O.bits_equal(got[1], ref[1])
                    for a, b in zip(got[2:], ref[2:]):
                        assert O.same_bits_or_nan(a, b)
        sc = scenes.get_scene("tris", n=71)
        frames = {}
        for bvh_min in ("1", "1000000000"):
            os.environ["RT_TRI_BVH_MIN"] = bvh_min
            rt.upload_scene(sc)
            frames[bvh_min], rays = rt.render(480, 270, 2, 8, seed=0xBEEF, flags=4)
            assert rt.verify_mismatches() == 0 and rays > 480 * 270 * 2
        assert O.same_bits_or_nan(frames["1"], frames["1000000000"])
        big = scenes.get_scene("tris", n=200)              # 79k triangles: two item slots per leaf
        os.environ["RT_TRI_BVH_MIN"] = "1"
        rt.upload_scene(big)
        _, rays = rt.render(480, 270, 1, 8, seed=0xBEEF, flags=4)
        assert rt.verify_mismatches() == 0 and rays > 480 * 270
    finally:
        os.environ.pop("RT_TRI_BVH_MIN", None)


@pytest.mark.parametrize("W,H", [(320, 180), (250, 37)])
def test_render_pixels_long_lists_are_sorted_by_tile_first(rt, W, H):
    """Lists of 2048 pixels and more are sorted by 8x8 tile on the device before they are rendered (random order costs the
    485-sphere frame 43 %, tools/pixels_order.py). The sort must not show: every listed pixel gets the frame's sums, a pixel
    listed k times gets k times them under RT_RENDER_ACCUMULATE, indices outside the frame are skipped, nothing else is
    written, the caller's list is left as it was, and a second call with a shorter list reuses the workspace."""
    from ray_amd import api
    sc = scenes.get_scene("random", half=3)
    rt.upload_scene(sc)
    spp, depth = 2, 8
    full, _ = rt.render(W, H, spp, depth, seed=77)
    f = full.reshape(-1, 4)
    rng = np.random.default_rng(W)
    for n in (30000, 2048, 5000):
        px = rng.integers(0, W * H, n).astype(np.int32)
        px[rng.integers(0, n, 40)] = rng.choice(np.array([-1, W * H, W * H + 3, 2 ** 31 - 1, -2 ** 31], np.int32), 40)
        valid = px[(px >= 0) & (px < W * H)]
        counts = np.bincount(valid, minlength=W * H)
        dpx = rt.alloc(px.nbytes).from_numpy(px)
        acc = rt.alloc((W * H + 16) * 16).zero()
        rays = rt.alloc(8).zero()
        rt.render_pixels(W, H, acc, dpx, spp, depth, seed=77, rays=rays)
        got = acc.to_numpy(np.float32, (W * H + 16, 4))
        assert O.same_bits_or_nan(got[:W * H][counts > 0], f[counts > 0])
        assert (got[:W * H][counts == 0] == 0).all() and (got[W * H:] == 0).all()
        assert np.array_equal(dpx.to_numpy(np.int32, (n,)), px)
        acc.zero()
        rt.render_pixels(W, H, acc, dpx, spp, depth, seed=77, flags=api.RENDER_ACCUMULATE)
        got = acc.to_numpy(np.float32, (W * H + 16, 4))
        want = np.zeros((W * H, 4), np.float32)
        for k in range(1, int(counts.max()) + 1):
            sel = counts >= k
            want[sel] = want[sel] + f[sel]
        assert O.same_bits_or_nan(got[:W * H], want) and (got[W * H:] == 0).all()
        acc.free(); dpx.free(); rays.free()


def test_sample_slices_give_the_same_frame(rt):
    """render_kernel hands a pixel's samples out as consecutive slices: a lane that takes a pixel of a later
    slice continues the sum the earlier slices left in the accumulator, in sample order, and waits for them through a per-tile
    counter. Whatever the slices, the accumulators, the ray count and the oracle's frame are the same bits. Ragged frame (partial
    tiles), sample counts that are no multiple of the unit, units of 1 sample on a frame with far fewer tiles than waves (every
    wave of a later slice waits for the earlier one), RT_RENDER_CONTINUE across calls, ACCUMULATE (one slice), bands."""
    from ray_amd import api
    sc = scenes.get_scene("random", half=3)
    rt.upload_scene(sc)
    W, H, spp, depth = 100, 45, 20, 8
    oacc, orays = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=77, math_mode=1, n_threads=8)
    try:
        # a frame with fewer tiles than waves goes through the colour buffer of the frame's end entirely (rt_set_tail_limit) ...
        acc, rays = rt.render(W, H, spp, depth, seed=77)
        assert rt.last_kernel_ms()[1] == 2 and rays == orays and O.same_bits_or_nan(acc, oacc)
        # ... unless the limit leaves room for some of its 78 tiles only (1 MiB: 51 tiles of 20 KiB), or for none: then slices
        for mb in (1, 0):
            rt.set_tail_limit(mb)
            for smin, smax in [(0, 16), (8, 16), (1, 16), (1, 3), (3, 16), (7, 2), (10, 16), (16, 16), (2, 1)]:
                rt.set_sample_slices(smin, smax)
                acc, rays = rt.render(W, H, spp, depth, seed=77)
                assert rt.last_kernel_ms()[1] == (2 if mb else 1)
                assert rays == orays and O.same_bits_or_nan(acc, oacc), (mb, smin, smax)
        rt.set_sample_slices(2, 16)
        # consecutive sample ranges continued in one accumulator: the bits of the one call, whatever was there before the first
        buf = rt.alloc(W * H * 16).from_numpy(np.full((H, W, 4), 7.5, np.float32))
        n = 0
        r = rt.alloc(8).zero()
        for s0, cnt in ((0, 7), (7, 5), (12, 0), (12, 8)):
            rt.render_into(W, H, buf, cnt, depth, seed=77, spp_begin=s0, flags=api.RENDER_CONTINUE if s0 else 0, rays=r)
        assert int(r.to_numpy(np.uint64, (1,))[0]) == orays
        assert O.same_bits_or_nan(buf.to_numpy(np.float32, (H, W, 4)), oacc)
        # a second pass ADDED to the first: old + (the pass's own sum)
        rt.render_into(W, H, buf, spp, depth, seed=77, flags=api.RENDER_ACCUMULATE)
        assert O.same_bits_or_nan(buf.to_numpy(np.float32, (H, W, 4)), oacc + oacc)
        assert rt.lib.rt_render(W, H, 0, H, 0, 4, depth, 77, api.RENDER_ACCUMULATE | api.RENDER_CONTINUE, buf.ptr, None) != 0
        buf.free(); r.free()
        # interleaved bands and a row range that starts inside a tile row, each with its own slices
        total = np.zeros_like(oacc)
        n = 0
        for k in range(3):
            part, nr = rt.render(W, H, spp, depth, seed=77, band_first=k, band_stride=3)
            total += part
            n += nr
        assert n == orays and O.same_bits_or_nan(total, oacc)
        part, _ = rt.render(W, H, spp, depth, seed=77, y_begin=5, y_count=23)
        assert O.same_bits_or_nan(part[5:28], oacc[5:28]) and (part[:5] == 0).all() and (part[28:] == 0).all()
        # the same with some tiles in the colour buffer: continued calls, bands
        rt.set_tail_limit(1)
        buf = rt.alloc(W * H * 16).zero()
        for s0, cnt in ((0, 16), (16, 4)):       # (the second call has too few samples for a tail)
            rt.render_into(W, H, buf, cnt, depth, seed=77, spp_begin=s0, flags=api.RENDER_CONTINUE if s0 else 0)
        assert O.same_bits_or_nan(buf.to_numpy(np.float32, (H, W, 4)), oacc)
        buf.zero()
        for s0, cnt in ((0, 2), (2, 18)):        # (the second call continues the first THROUGH the tail's sum kernel)
            rt.render_into(W, H, buf, cnt, depth, seed=77, spp_begin=s0, flags=api.RENDER_CONTINUE if s0 else 0)
            assert rt.last_kernel_ms()[1] == (2 if s0 else 1)
        assert O.same_bits_or_nan(buf.to_numpy(np.float32, (H, W, 4)), oacc)
        rt.render_into(W, H, buf, spp, depth, seed=77, flags=api.RENDER_ACCUMULATE)
        assert O.same_bits_or_nan(buf.to_numpy(np.float32, (H, W, 4)), oacc + oacc)
        buf.free()
        total = np.zeros_like(oacc)
        for k in range(3):
            part, nr = rt.render(W, H, spp, depth, seed=77, band_first=k, band_stride=3)
            total += part
        assert O.same_bits_or_nan(total, oacc)
    finally:
        rt.set_sample_slices(8, 16)
        rt.set_tail_limit(1400)
    # a frame with several tiles per wave: the three ways to render it
    sc2 = scenes.builtin_scene()
    rt.upload_scene(sc2)
    a, ra = rt.render(1280, 720, 32, 10, seed=5)          # (whole pixels + half the frame through the colour buffer)
    rt.set_tail_limit(16)
    try:
        c, rc = rt.render(1280, 720, 32, 10, seed=5)      # (the limit cuts the tail short: sample slices + 512 tail tiles)
        assert rt.last_kernel_ms()[1] == 2 and rc == ra and O.same_bits_or_nan(a, c)
        rt.set_sample_slices(0, 16)
        rt.set_tail_limit(0)
        b, rb = rt.render(1280, 720, 32, 10, seed=5)      # (whole pixels only)
        assert rt.last_kernel_ms()[1] == 1
    finally:
        rt.set_sample_slices(8, 16)
        rt.set_tail_limit(1400)
    assert ra == rb and O.same_bits_or_nan(a, b)


def test_page_locked_host_memory_in_place_of_device_buffers(rt):
    """rt_malloc_host: rays read from, and colours / hit records written to, page-locked host memory by the kernels themselves (what
    the reference-signature calc::ray_color / hitFromList of the mirror headers use for their one ray): the same bits as through
    device buffers and copies."""
    import ctypes as C
    sc = scenes.builtin_scene()
    rt.upload_scene(sc)
    n = 777
    rng = np.random.default_rng(3)
    ro = rng.uniform(-3, 3, (n, 3)).astype(np.float32); ro[:, 2] += 1.0
    rd = rng.standard_normal((n, 3)).astype(np.float32)
    want = rt.trace_rays(ro, rd, 10, seed=42)
    whit = rt.hit_rays(ro, rd, 0.001, np.inf)
    p = C.c_void_p()
    nbytes = n * (24 + 12 + 4 + 8 + 12 + 12 + 1) + 64
    assert rt.lib.rt_malloc_host(C.byref(p), nbytes) == 0, rt.lib.rt_last_error()
    try:
        base = p.value
        host = (C.c_char * nbytes).from_address(base)
        view = np.frombuffer(host, np.uint8)
        off = {"o": 0, "d": 12 * n, "rgb": 24 * n, "t": 36 * n + (-(36 * n) % 8)}
        off["prim"] = off["t"] + 8 * n; off["p"] = off["prim"] + 4 * n; off["nrm"] = off["p"] + 12 * n; off["ff"] = off["nrm"] + 12 * n
        view[off["o"]:off["o"] + 12 * n] = ro.view(np.uint8).reshape(-1)
        view[off["d"]:off["d"] + 12 * n] = rd.view(np.uint8).reshape(-1)
        assert rt.lib.rt_trace_rays(n, base + off["o"], base + off["d"], 10, 42, 0, base + off["rgb"]) == 0, rt.lib.rt_last_error()
        got = view[off["rgb"]:off["rgb"] + 12 * n].view(np.float32).reshape(n, 3)
        assert O.same_bits_or_nan(got, want)
        assert rt.lib.rt_hit_rays(n, base + off["o"], base + off["d"], 0.001, float("inf"), base + off["prim"], base + off["t"], base + off["p"],
                                  base + off["nrm"], base + off["ff"]) == 0, rt.lib.rt_last_error()
        assert np.array_equal(view[off["prim"]:off["prim"] + 4 * n].view(np.int32), whit[0])
        assert O.bits_equal(view[off["t"]:off["t"] + 8 * n].view(np.float64), whit[1])
        assert O.same_bits_or_nan(view[off["p"]:off["p"] + 12 * n].view(np.float32).reshape(n, 3), whit[2])
        assert O.same_bits_or_nan(view[off["nrm"]:off["nrm"] + 12 * n].view(np.float32).reshape(n, 3), whit[3])
        assert np.array_equal(view[off["ff"]:off["ff"] + n], whit[4].view(np.uint8).reshape(-1))
    finally:
        assert rt.lib.rt_free_host(p) == 0


def test_block_sorted_shading_gives_the_same_frame(rt):
    """rt_set_shading_sort(1): the bounces of a block's rays shaded in order of material by whichever thread holds the
    slot, weights pushed on the owner's path stack. Same accumulators and ray counts as the oracle / the default kernel: the built-in
    scene at depth 10 (levels beyond the LDS part of the stack), a ragged frame with fewer tiles than waves (blocks with idle waves),
    sample slices without a colour buffer, bands, the normal view, a random scene with a filter group (the barriers that keep scans and
    slots apart), the Cornell-style box scene (many rects: the mode is ignored)."""
    from ray_amd import api
    try:
        sc = scenes.builtin_scene()
        rt.upload_scene(sc)
        W, H, spp, depth = 100, 45, 20, 10
        oacc, orays = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=5, math_mode=1, n_threads=8)
        rt.set_shading_sort(1)
        acc, rays = rt.render(W, H, spp, depth, seed=5)
        assert rays == orays and O.same_bits_or_nan(acc, oacc)
        rt.set_tail_limit(0)
        rt.set_sample_slices(2, 16)
        acc, rays = rt.render(W, H, spp, depth, seed=5)
        assert rt.last_kernel_ms()[1] == 1 and rays == orays and O.same_bits_or_nan(acc, oacc)
        rt.set_tail_limit(1400)
        rt.set_sample_slices(8, 16)
        nacc, _ = O.render(sc, W, H, 2, depth, rng_mode=1, seed64=5, math_mode=1, n_threads=8, flags=api.RENDER_NORMALS)
        acc, _ = rt.render(W, H, 2, depth, seed=5, flags=api.RENDER_NORMALS)
        assert O.same_bits_or_nan(acc, nacc)
        for name, kw, w, h, n in (("builtin", {}, 1280, 720, 16), ("random", {"half": 5}, 320, 200, 8), ("box", {}, 160, 120, 4)):
            rt.upload_scene(scenes.get_scene(name, **kw))
            got = {}
            for mode in (0, 1):
                rt.set_shading_sort(mode)
                buf = rt.alloc(w * h * 16).zero()
                r = rt.alloc(8).zero()
                rt.render_into(w, h, buf, n, 8, seed=9, rays=r, band_first=1, band_stride=3)
                got[mode] = (buf.to_numpy(np.uint32, (h, w, 4)).copy(), int(r.to_numpy(np.uint64, (1,))[0]))
                buf.free(); r.free()
            assert got[0][1] == got[1][1] and np.array_equal(got[0][0], got[1][0]), name
    finally:
        rt.set_shading_sort(0)
        rt.set_tail_limit(1400)
        rt.set_sample_slices(8, 16)


def test_triangle_queue_overflow_negative_t_and_zero_t(rt):
    """scan_triangles shares the flagged items of a wave's rays out among its lanes through a queue in LDS and folds hits into the
    owner's cell with an atomic min on (t, ~index). Stress what the frames do not: rays through 20 stacked layers of a dense
    mesh (hundreds of entries per wave: the queue fills up and is worked off in the middle of a chunk), a negative t_min (hits
    behind the origin: keys of negative floats), origins exactly on a layer (t == 0: those rays take the list-order scan), a
    ray count that leaves the last wave half empty, and ties between coplanar duplicates (the later primitive wins)."""
    rng = np.random.default_rng(8)
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    shapes = []
    n, layers = 10, 20
    for L in range(layers):
        z = np.float32(L) * np.float32(0.5)
        for i in range(n):
            for j in range(n):
                a, b, c, d = (i, j, z), (i + 1, j, z), (i + 1, j + 1, z), (i, j + 1, z)
                shapes.append(scenes.Shape("triangle", tuple(float(v) for v in (*a, *b, *c)), 0))
                shapes.append(scenes.Shape("triangle", tuple(float(v) for v in (*a, *c, *d)), 0))
    for k in range(30):            # duplicates of some triangles of layer 3: equal t, the later index must win
        s = shapes[3 * 2 * n * n + 7 * k]
        shapes.append(scenes.Shape("triangle", s.data, 0))
    sc = scenes.Scene(mats, shapes, None, scenes.rayview_camera())
    rt.upload_scene(sc, set_camera=False)
    m = 64 * 37 + 29
    ro = np.empty((m, 3), np.float32)
    ro[:, 0] = rng.uniform(0.05, n - 0.05, m); ro[:, 1] = rng.uniform(0.05, n - 0.05, m)
    zsel = rng.integers(0, 3, m)
    ro[:, 2] = np.where(zsel == 0, -2.0, np.where(zsel == 1, rng.integers(0, layers, m) * 0.5, rng.uniform(-1, layers * 0.5, m))).astype(np.float32)
    rd = np.empty((m, 3), np.float32)
    rd[:, 0] = rng.uniform(-0.02, 0.02, m); rd[:, 1] = rng.uniform(-0.02, 0.02, m)
    rd[:, 2] = np.where(rng.random(m) < 0.5, 1.0, -1.0) * rng.uniform(0.5, 2.0, m)
    for t_min, t_max in ((1e-4, np.inf), (-100.0, np.inf), (-3.0, 2.5), (0.0, np.inf)):
        got = rt.hit_rays(ro, rd, t_min, t_max)
        ref = O.hit_rays(sc, ro, rd, t_min, t_max)
        assert np.array_equal(got[0], ref[0]), (t_min, t_max)
        assert O.bits_equal(got[1], ref[1]), (t_min, t_max)
        assert (got[0] >= 0).mean() > 0.6
    assert (ref[1][ref[0] >= 0] == 0).any()                       # the t == 0 path was taken
    neg = rt.hit_rays(ro, rd, -100.0, np.inf)
    assert (neg[1][neg[0] >= 0] < 0).mean() > 0.3                # and hits behind the origin are found
    dup = rt.hit_rays(ro, rd, 1e-4, np.inf)[0]
    assert (dup >= 2 * n * n * layers).any()                     # a duplicate (later index) won a tie
