"""Pins the CPU oracle (oracle/rt_oracle.cpp) against golden vectors produced by the reference's
own code (oracle/_ref/ref_harness, see oracle/make_golden.py). Everything here is bit-exact unless
a tolerance is written in the test. CPU only."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import oracle_lib as O
from oracle_lib import P
from ray_amd import scenes
from oracle.make_golden import golden_scenes

f4, f8, i4, u1 = C.c_float, C.c_double, C.c_int32, C.c_uint8


# ---- RNG (calc.h:103-113, calc.cpp:12-72, calc.h:62-86) ------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2, 12345])
def test_refrng_streams(seed):
    g = O.golden("rng")
    n = 4096
    out = np.empty(n, np.float32)
    O.lib().orc_refrng_fill(seed, 0, 0, 1, 0, n, P(out, f4)); assert O.bits_equal(out, g[f"seed{seed}_u01"])
    O.lib().orc_refrng_fill(seed, 1, -1, 1, 0, n, P(out, f4)); assert O.bits_equal(out, g[f"seed{seed}_u11"])
    O.lib().orc_refrng_fill(seed, 2, 0, 1, 0, n, P(out, f4)); assert O.bits_equal(out, g[f"seed{seed}_gen_0_1"])
    O.lib().orc_refrng_fill(seed, 2, 3, 5, n, n, P(out, f4)); assert O.bits_equal(out, g[f"seed{seed}_gen_3_5"])


def test_derived_samplers():
    g = O.golden("rng")
    n = 512
    o = [np.empty((n, 3), np.float32) for _ in range(4)]
    O.lib().orc_samplers(7, 8, 9, 0, n, 90.0, 40000.0, 100.0, *[P(a, f4) for a in o])
    assert O.bits_equal(o[0], g["in_unit_sphere_s9"])
    assert O.bits_equal(o[1], g["in_unit_disk_s9"])
    assert O.bits_equal(o[2], g["cosine_direction_s8"])
    assert O.bits_equal(o[3], g["to_sphere_s8"])


def test_counter_rng_matches_python_twin():
    out = np.empty(64, np.float32)
    for seed, px, s in [(0, 0, 0), (0xC0FFEE, 12345, 7), ((5 << 32) | 9, 2073599, 255)]:
        O.lib().orc_counter_fill(seed, px, s, 64, P(out, f4))
        r = scenes.CounterRng(seed, px, s)
        ref = np.array([r.u01() for _ in range(64)], np.float32)
        assert O.bits_equal(out, ref)
        assert (out >= 0).all() and (out < 1).all()


# ---- QVector3D (Qt5Gui) ----------------------------------------------------------------------------
def test_vec3_ops():
    g = O.golden("vec3")
    a, b = g["a"], g["b"]
    n = a.shape[0]
    lsq, ln, dt = (np.empty(n, np.float32) for _ in range(3))
    nrm, crs = (np.empty((n, 3), np.float32) for _ in range(2))
    O.lib().orc_vec3(n, P(a, f4), P(b, f4), P(lsq, f4), P(ln, f4), P(nrm, f4), P(dt, f4), P(crs, f4))
    assert O.bits_equal(lsq, g["lengthSquared"])
    assert O.bits_equal(ln, g["length"])
    assert O.bits_equal(nrm, g["normalized"])
    assert O.bits_equal(dt, g["dot"])
    assert O.bits_equal(crs, g["cross"])


# ---- camera (camera.h) -----------------------------------------------------------------------------
def test_camera_frame_and_rays():
    g = O.golden("camera")
    vecs, prm = g["cfg_vecs"], g["cfg_params"]
    for i in range(vecs.shape[0]):
        cam = O.RtCamera()
        v = [np.ascontiguousarray(vecs[i, k]) for k in range(3)]
        O.lib().orc_camera(P(v[0], f4), P(v[1], f4), P(v[2], f4), *[float(x) for x in prm[i]], C.byref(cam))
        assert O.bits_equal(cam.as_array(), g[f"cam{i}_frame"])
        assert cam.lens_radius == g[f"cam{i}_lens_radius"][0]
        st = np.ascontiguousarray(g[f"cam{i}_st"])
        n = st.shape[0]
        ro, rd = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        dl2 = np.empty(n, np.float32)
        O.lib().orc_camera_rays(C.byref(cam), 11, 12, int(g[f"cam{i}_seed11"][0]), n, P(st, f4), P(ro, f4), P(rd, f4), P(dl2, f4))
        assert O.bits_equal(ro, g[f"cam{i}_ray_o"])
        assert O.bits_equal(rd, g[f"cam{i}_ray_d"])
        assert O.bits_equal(dl2, g[f"cam{i}_ray_dl2"])


# ---- sphere::hit / rect hits (shapes.h) --------------------------------------------------------------
def _shape_hits(kind, prm9, ro, rd, tmax):
    n = ro.shape[0]
    has, ff = np.empty(n, np.uint8), np.empty(n, np.uint8)
    t, uv = np.empty(n, np.float64), np.empty((n, 2), np.float64)
    p, nrm = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    O.lib().orc_shape_hits(n, P(kind, i4), P(prm9, f8), P(np.ascontiguousarray(ro), f4), P(np.ascontiguousarray(rd), f4),
                           0.001, P(np.ascontiguousarray(tmax), f8), P(has, u1), P(t, f8), P(p, f4), P(nrm, f4), P(ff, u1), P(uv, f8))
    return has, t, p, nrm, ff, uv


def _check_hits(got, g, pre):
    has, t, p, nrm, ff, uv = got
    assert np.array_equal(has, g[pre + "_has"])
    assert 0.05 < has.mean() < 0.95                      # the fixture exercises both outcomes
    assert O.same_values(t, g[pre + "_t"]) and O.bits_equal(np.nan_to_num(t), np.nan_to_num(g[pre + "_t"]))
    assert O.same_values(p, g[pre + "_p"])
    assert O.same_values(nrm, g[pre + "_n"])
    assert np.array_equal(ff, g[pre + "_ff"])
    assert O.same_values(uv, g[pre + "_uv"])


def test_sphere_hits():
    g = O.golden("shapehits")
    n = g["sph_prm"].shape[0]
    prm9 = np.zeros((n, 9), np.float64); prm9[:, :4] = g["sph_prm"]
    got = _shape_hits(np.zeros(n, np.int32), prm9, g["sph_ray_o"], g["sph_ray_d"], g["sph_tmax"])
    _check_hits(got, g, "sph")


def test_rect_hits_including_parallel_rays():
    g = O.golden("shapehits")
    n = g["rect_prm"].shape[0]
    prm9 = np.zeros((n, 9), np.float64); prm9[:, :5] = g["rect_prm"]
    kind = (g["rect_axis"] + 1).astype(np.int32)
    got = _shape_hits(kind, prm9, g["rect_ray_o"], g["rect_ray_d"], g["rect_tmax"])
    _check_hits(got, g, "rect")


# ---- materials / pdfs (hitPosition.h, hitPosition.cpp:55-72, shapes.h pdf_value/random) -------------
def _materials(math_mode):
    g = O.golden("materials")
    n = g["in_d"].shape[0]
    k = O.OrcMatKat()
    k.n, k.math_mode = n, math_mode
    keep = []

    def inp(name, arr, ty):
        a = np.ascontiguousarray(arr); keep.append(a); setattr(k, name, P(a, ty))

    inp("in_d", g["in_d"], f4); inp("in_n", g["in_n"], f4); inp("in_p", g["in_p"], f4); inp("in_ff", g["in_ff"], u1)
    inp("prm", g["prm"], f4); inp("ir", g["ir"], f8); inp("lamb_sd", g["lamb_sd"], f4)
    inp("xz_o", g["xzlight_gen_and_o"][:, 1, :], f4)
    out = {}
    for name, shape, dt, ty in [("metal_o", (n, 3), np.float32, f4), ("metal_d", (n, 3), np.float32, f4),
                                ("metal_att", (n, 3), np.float32, f4), ("diel_d", (n, 3), np.float32, f4),
                                ("diel_att", (n, 3), np.float32, f4), ("lamb_att", (n, 3), np.float32, f4),
                                ("lamb_spdf", (n,), np.float64, f8), ("cos_val", (n,), np.float64, f8),
                                ("cos_gen", (n, 3), np.float32, f4), ("sph_gen", (n, 3), np.float32, f4),
                                ("sph_val", (n,), np.float64, f8), ("xz_gen", (n, 3), np.float32, f4),
                                ("xz_val", (n,), np.float64, f8), ("xy_gen", (n, 3), np.float32, f4),
                                ("xy_val", (n,), np.float64, f8), ("yz_gen", (n, 3), np.float32, f4),
                                ("yz_val", (n,), np.float64, f8), ("mix_gen", (n, 3), np.float32, f4),
                                ("mix_val", (n,), np.float64, f8), ("refl", (n, 3), np.float32, f4),
                                ("refr", (n, 3), np.float32, f4), ("refr_eta", (n,), np.float64, f8)]:
        out[name] = np.empty(shape, dt); setattr(k, name, P(out[name], ty))
    O.lib().orc_materials_kat(C.byref(k))
    return g, out


_MAT_KEYS = {"metal_o": "metal_o", "metal_d": "metal_d", "metal_att": "metal_att", "diel_d": "diel_d",
             "diel_att": "diel_att", "lamb_att": "lamb_att", "lamb_spdf": "lamb_spdf", "cos_val": "cos_val",
             "cos_gen": "cos_gen", "sph_gen": "sphlight_gen", "sph_val": "sphlight_val", "xz_val": "xzlight_val",
             "xy_gen": "xylight_gen", "xy_val": "xylight_val", "yz_gen": "yzlight_gen", "yz_val": "yzlight_val",
             "mix_gen": "mix_gen", "mix_val": "mix_val", "refl": "reflect", "refr": "refract", "refr_eta": "refract_eta"}


def test_materials_and_pdfs_bit_exact():
    g, out = _materials(math_mode=0)
    for mine, theirs in _MAT_KEYS.items():
        assert O.same_values(out[mine], g[theirs]), mine
    assert O.same_values(out["xz_gen"], g["xzlight_gen_and_o"][:, 0, :])
    # the fixture covers reflection, refraction and total internal reflection
    assert (g["yzlight_val"] == 0).all() and (g["yzlight_gen"] == np.array([1, 0, 0], np.float32)).all()
    assert (g["xylight_val"] > 0).any() and (g["sphlight_val"] > 0).any() and (g["xzlight_val"] > 0).any()


def test_materials_portable_math_close_to_libm():
    """math_mode 1 (portable sin/cos/pow5, shared with the GPU) vs the reference's libm: the RNG
    streams stay in step, so outputs agree to FP32 rounding: <= 1 ulp(float) relative 1.2e-7 on
    generated directions, and exactly on everything that uses no transcendental."""
    g, out = _materials(math_mode=1)
    for mine in ("metal_o", "metal_d", "lamb_spdf", "cos_val", "refl", "refr", "xz_gen", "xy_gen"):
        theirs = _MAT_KEYS.get(mine, None)
        ref = g["xzlight_gen_and_o"][:, 0, :] if mine == "xz_gen" else g[theirs]
        assert O.same_values(out[mine], ref), mine
    for mine in ("cos_gen", "sph_gen", "mix_gen", "diel_d"):
        ref = g[_MAT_KEYS[mine]]
        err = np.abs(out[mine].astype(np.float64) - ref.astype(np.float64))
        scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-30)
        assert (err / scale).max() <= 2.5e-7, mine
        assert (out[mine] == ref).mean() > 0.99, mine


def test_tonemap():
    g = O.golden("materials")
    tin, ns = np.ascontiguousarray(g["tone_in"]), np.ascontiguousarray(g["tone_n"])
    out = np.empty_like(tin)
    O.lib().orc_tonemap(tin.shape[0], P(tin, f4), P(ns, i4), P(out, f4))
    assert O.bits_equal(out, g["tone_out"])


# ---- portable transcendentals ---------------------------------------------------------------------
def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def test_portable_sincos_within_1ulp_of_libm():
    rng = np.random.default_rng(5)
    r1 = rng.integers(0, 1 << 24, 200000).astype(np.float32) * np.float32(2.0 ** -24)
    x = np.concatenate([2 * 3.1415926535897932385 * r1.astype(np.float64), np.linspace(0, 2 * np.pi, 10001)])
    s0, c0, s1, c1 = (np.empty_like(x) for _ in range(4))
    O.lib().orc_sincos(0, x.size, P(x, f8), P(s0, f8), P(c0, f8))
    O.lib().orc_sincos(1, x.size, P(x, f8), P(s1, f8), P(c1, f8))
    # 1 ulp where the value is not tiny; next to the zeros (x near k*pi/2) the two-term pi/2
    # reduction bounds the ABSOLUTE error instead (the reference narrows these values to float)
    big_s, big_c = np.abs(s0) > 1e-3, np.abs(c0) > 1e-3
    assert _ulps(s0, s1)[big_s].max() <= 1 and _ulps(c0, c1)[big_c].max() <= 1
    assert np.abs(s0 - s1).max() <= 1.2e-16 and np.abs(c0 - c1).max() <= 1.2e-16
    assert (s0 == s1).mean() > 0.9 and (c0 == c1).mean() > 0.9
    # after the reference's narrowing to float almost every value is identical
    assert (s0.astype(np.float32) == s1.astype(np.float32)).mean() > 0.999999 - 1e-4


def test_portable_pow5_within_4ulp_of_libm():
    x = np.random.default_rng(6).random(100000)
    y0, y1 = np.empty_like(x), np.empty_like(x)
    O.lib().orc_pow5(0, x.size, P(x, f8), P(y0, f8)); O.lib().orc_pow5(1, x.size, P(x, f8), P(y1, f8))
    # three rounded multiplications vs a (nearly) correctly rounded pow: a few double ulps, which
    # disappear in the reference's (float) cast of the result (hitPosition.h:286)
    assert _ulps(y0, y1).max() <= 4
    assert (y0.astype(np.float32) == y1.astype(np.float32)).mean() > 0.99999


# ---- hitFromList over whole frames: primary-hit ids (hitPosition.cpp:6-22, calc.cpp:99) -------------
@pytest.mark.parametrize("sname,W,H", [("builtin_a0", 640, 360), ("random11_a0", 640, 360), ("box_a0", 640, 360),
                                       ("builtin_a0", 1920, 1080)])
def test_primary_hit_ids_bit_exact(sname, W, H):
    g = O.golden(f"primary_{sname}_{W}x{H}")
    sc = golden_scenes()[sname]
    flat = sc.flatten()
    prim, t = O.primary(sc, W, H, rng_mode=0)
    outer = np.where(prim >= 0, flat["prim_outer"][np.maximum(prim, 0)], -1)
    side = np.where(prim >= 0, flat["prim_side"][np.maximum(prim, 0)], -1)
    assert np.array_equal(outer, g["prim_id"])
    assert np.array_equal(side, g["box_side"])
    assert O.bits_equal(t[::9, ::9].copy(), g["t_sub"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(t.tobytes()).digest(), np.uint8), g["t_sha256"])


# ---- calc::ray_color, sequence-exact with the reference's seeded engines (calc.cpp:90-131) ---------
@pytest.mark.parametrize("sname,W,H,spp,depth", [("builtin", 64, 36, 4, 8), ("random3", 48, 27, 2, 10),
                                                 ("box_a005", 48, 27, 2, 6), ("box_nanlight", 32, 18, 2, 6),
                                                 ("builtin_xylight", 48, 27, 4, 8),     # lights = an XY `rectangle` (quirk Q2)
                                                 ("builtin_yzlight", 48, 27, 4, 8)])    # lights = a yz_rect: base-class pdf
def test_seeded_radiance_sums_bit_exact(sname, W, H, spp, depth):
    g = O.golden(f"render_{sname}_{W}x{H}_s{spp}_d{depth}")
    sc = golden_scenes()[sname]
    acc, rays = O.render(sc, W, H, spp, depth, rng_mode=0, seeds=(1, 2, 3), math_mode=0)
    assert O.same_values(acc[..., :3], g["sum"])
    assert (acc[..., 3] == spp).all() and rays >= W * H * spp
    if sname == "box_nanlight":
        assert np.isnan(g["sum"]).any()        # the NaN-accepting comparisons are exercised


def test_counter_rng_portable_render_converges_to_reference_mean():
    """Oracle in the configuration the GPU is compared with (CounterRng + portable math) against the
    reference's 4096-spp converged image: same estimator, different random numbers. Tolerance: the
    mean over the frame within 1 %, and the RMS pixel difference consistent with 256-spp noise."""
    g = O.golden("mean_builtin_160x90")
    ref = g["sum"] / 4096.0
    sc = golden_scenes()["builtin"]
    acc, _ = O.render(sc, 160, 90, 256, 8, rng_mode=1, seed64=0xC0FFEE, math_mode=1, n_threads=8)
    img = acc[..., :3] / 256.0
    assert abs(img.mean() - ref.mean()) / ref.mean() < 0.01
    rms = np.sqrt(np.mean((img - ref) ** 2))
    assert rms < 0.05


def test_render_split_invariance():
    """Any split of the sample range / rows reproduces the same per-sample values (CounterRng)."""
    sc = golden_scenes()["builtin"]
    W, H = 32, 18
    full, rays_full = O.render(sc, W, H, 4, 8, rng_mode=1, seed64=7, math_mode=1)
    a, r1 = O.render(sc, W, H, 2, 8, rng_mode=1, seed64=7, math_mode=1, spp_begin=0)
    b, r2 = O.render(sc, W, H, 2, 8, rng_mode=1, seed64=7, math_mode=1, spp_begin=2)
    assert r1 + r2 == rays_full
    np.testing.assert_allclose(a[..., :3] + b[..., :3], full[..., :3], rtol=1e-6, atol=1e-6)
    top, _ = O.render(sc, W, H, 4, 8, rng_mode=1, seed64=7, math_mode=1, y_begin=0, y_count=9)
    bot, _ = O.render(sc, W, H, 4, 8, rng_mode=1, seed64=7, math_mode=1, y_begin=9, y_count=9, accum=top)
    assert O.same_values(bot, full)
    thr, _ = O.render(sc, W, H, 4, 8, rng_mode=1, seed64=7, math_mode=1, n_threads=4)
    assert O.same_values(thr, full)


# ---- triangles (not in the reference): analytic known answers, tests/tri_kat.py ------------------------
def test_triangle_known_answers():
    """The oracle's Triangle::hit against hand-computed hits on the unit triangle (centre, edges, vertices, back
    face, parallel and in-plane rays, t-range ends), an independent float64 computation on random triangles, and
    the reference's `rectangle` tiled by two triangles."""
    import tri_kat
    tri_kat.check_unit_triangle(O.hit_rays)
    for seed in (5, 6):
        tri_kat.check_general_triangles(O.hit_rays, seed=seed)
    tri_kat.check_triangle_pair_equals_rectangle(O.hit_rays)
    tri_kat.check_t_min_that_is_no_float(O.hit_rays)


# ---- constant_medium::hit (shapes.h:374-455) ---------------------------------------------------------------
def _medium_kat(math_mode):
    g = O.golden("medium")
    n = g["kind"].shape[0]
    has, t, p = np.empty(n, np.uint8), np.empty(n), np.empty((n, 3), np.float32)
    nrm, ff = np.empty((n, 3), np.float32), np.empty(n, np.uint8)
    O.lib().orc_medium_kat(n, math_mode, O.P(g["kind"], C.c_int32), O.P(g["bound"], C.c_float), O.P(g["density"], C.c_double),
                           O.P(g["ray_o"], C.c_float), O.P(g["ray_d"], C.c_float), O.P(g["t_min"], C.c_double), O.P(g["t_max"], C.c_double),
                           O.P(has, C.c_uint8), O.P(t, C.c_double), O.P(p, C.c_float), O.P(nrm, C.c_float), O.P(ff, C.c_uint8))
    return g, has, t, p, nrm, ff


def test_constant_medium_hits_bit_exact():
    """6000 seeded cases from the reference (sphere and box boundaries, rays through / inside / missing / parallel to
    faces, densities 0.01..100, t ranges incl. negative t_min and finite t_max): has, t bits, p, normal, front_face."""
    g, has, t, p, nrm, ff = _medium_kat(0)
    assert np.array_equal(has, g["has"]) and 0.3 < has.mean() < 0.6
    assert O.bits_equal(t, g["t"]) and O.same_values(p, g["p"])
    assert np.array_equal(nrm, g["n"]) and np.array_equal(ff, g["ff"])
    assert (nrm[has == 1] == (1, 0, 0)).all() and (ff[has == 1] == 1).all()          # "arbitrary", shapes.h:439-440


def test_portable_log_within_one_ulp_of_libm():
    """math_mode 1 replaces libm's log (not reproducible on the GPU) by fdlibm's algorithm in IEEE operations: never
    more than 1 ulp from libm on every value a float draw can take, and the medium fixture moves by at most that."""
    x = np.concatenate([np.arange(1, 2 ** 24, 37, dtype=np.float64) * 2.0 ** -24, np.random.default_rng(0).random(100000)])
    y0, y1 = np.empty_like(x), np.empty_like(x)
    O.lib().orc_log(0, x.size, O.P(x, C.c_double), O.P(y0, C.c_double))
    O.lib().orc_log(1, x.size, O.P(x, C.c_double), O.P(y1, C.c_double))
    assert _ulps(y0, y1).max() <= 1
    z = np.zeros(1); out = np.empty(1)
    O.lib().orc_log(1, 1, O.P(z, C.c_double), O.P(out, C.c_double))
    assert out[0] == -np.inf
    g, has, t, p, nrm, ff = _medium_kat(1)
    assert np.array_equal(has, g["has"])
    both = has == 1
    assert _ulps(t[both], g["t"][both]).max() <= 4 and (t != g["t"]).mean() < 0.02
