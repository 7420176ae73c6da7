"""The C++ mirror headers (ray_amd/host/{Ray,calc,hitPosition,shapes,camera}.h) keep the reference's signatures;
their HOST arithmetic (the engines and samplers of calc.h, QVector3D, camera::resetCam / get_ray, rgbPerSamples,
reflect / refract) is pinned here against the goldens that the compiled reference produced, bit for bit, by running
the same call sequences through ray_amd/host/mirror_kat (no GPU needed: nothing here touches the device)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = os.path.join(ROOT, "ray_amd", "host", "mirror_kat")


@pytest.fixture(scope="module")
def kat(tmp_path_factory):
    if not os.path.exists(KAT):
        import __graft_entry__ as ge
        ge.build()
    tmp = tmp_path_factory.mktemp("kat")

    def run(*args, inputs=None):
        paths = []
        if inputs is not None:
            p = tmp / "in.bin"
            with open(p, "wb") as f:
                for a in inputs:
                    f.write(np.ascontiguousarray(a).tobytes())
            paths.append(str(p))
        out = tmp / "out.bin"
        subprocess.run([KAT] + [str(a) for a in args] + paths + [str(out)], check=True, timeout=60)
        return np.fromfile(out, np.uint8)
    return run


def _take(buf, pos, dtype, shape):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return buf[pos:pos + n].view(dtype).reshape(shape), pos + n


def test_engines_and_samplers_match_the_reference(kat):
    g = O.golden("rng")
    for seed in (0, 1, 2, 12345):
        b, pos = kat("rng", seed), 0
        for key, dt in (("u01", np.float32), ("u11", np.float32), ("gen_0_1", np.float32), ("gen_3_5", np.float32),
                        ("W", np.int32), ("H", np.int32)):
            got, pos = _take(b, pos, dt, (4096,))
            assert np.array_equal(got.view(np.uint32), g[f"seed{seed}_{key}"].view(np.uint32)), (seed, key)
    b, pos = kat("samplers"), 0
    for key in ("in_unit_sphere_s9", "in_unit_disk_s9", "cosine_direction_s8", "to_sphere_s8"):
        got, pos = _take(b, pos, np.float32, (512, 3))
        assert O.bits_equal(got, g[key]), key


def test_qvector3d_standin_matches_qt(kat):
    g = O.golden("vec3")
    n = g["a"].shape[0]
    b, pos = kat("vec3", n, inputs=[g["a"], g["b"]]), 0
    for key, shape in (("lengthSquared", (n,)), ("length", (n,)), ("normalized", (n, 3)), ("dot", (n,)), ("cross", (n, 3))):
        got, pos = _take(b, pos, np.float32, shape)
        assert O.same_values(got, g[key]), key


def test_camera_frame_and_get_ray_match_the_reference(kat):
    g = O.golden("camera")
    for i in range(g["cfg_vecs"].shape[0]):
        st = g[f"cam{i}_st"]
        b, pos = kat("camera", int(g[f"cam{i}_seed11"][0]), st.shape[0], inputs=[g["cfg_vecs"][i], g["cfg_params"][i], st]), 0
        frame, pos = _take(b, pos, np.float32, (7, 3))
        lens, pos = _take(b, pos, np.float64, (1,))
        assert O.bits_equal(frame, g[f"cam{i}_frame"]) and lens[0] == g[f"cam{i}_lens_radius"][0]
        for key, shape in (("ray_o", (st.shape[0], 3)), ("ray_d", (st.shape[0], 3)), ("ray_dl2", (st.shape[0],))):
            got, pos = _take(b, pos, np.float32, shape)
            assert O.bits_equal(got, g[f"cam{i}_{key}"]), (i, key)


def test_tonemap_and_optics_match_the_reference(kat):
    g = O.golden("materials")
    n = g["tone_in"].shape[0]
    got, _ = _take(kat("tone", n, inputs=[g["tone_in"], g["tone_n"]]), 0, np.float32, (n, 3))
    assert O.same_values(got, g["tone_out"])
    n = g["in_d"].shape[0]
    b = kat("optics", n, inputs=[g["in_d"], g["in_n"], g["refract_eta"]])
    refl, pos = _take(b, 0, np.float32, (n, 3))
    refr, pos = _take(b, pos, np.float32, (n, 3))
    assert O.same_values(refl, g["reflect"]) and O.same_values(refr, g["refract"])


def test_a_caller_shaped_like_renderOneRay_compiles_against_the_mirror_headers(tmp_path):
    """rayview_tick.cpp uses cam.get_ray(u, v), the six-argument calc::ray_color, the four-argument hitFromList,
    calc::rgbPerSamples and shape::hit / bounding_box with the reference's signatures and no session object; it must
    compile and link against ray_amd/host/*.h + libray_amd.so (running it needs the GPU: tests/test_gpu_parity.py)."""
    src = os.path.join(ROOT, "ray_amd", "host", "rayview_tick.cpp")
    text = open(src).read()
    assert "Session" not in text and "rt_init" not in text
    csrc = os.path.join(ROOT, "ray_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-Werror", "-I", os.path.join(ROOT, "ray_amd", "host"), "-I",
                        os.path.join(ROOT, "include"), src, "-o", str(tmp_path / "tick"), "-L", csrc, "-lray_amd", "-Wl,-rpath," + csrc],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_host_hit_arithmetic_and_uv_match_the_reference(kat):
    """shape::hit_host (what shape::pdf_value asks one shape about one ray) and hit_record::u, v (get_sphere_uv, the rects' (x - x0) /
    (x1 - x0)): the 48k sphere and rect cases of the compiled reference, incl. tangent, inside, on-surface, edge and axis-parallel rays."""
    g = O.golden("shapehits")
    for pre, kind, prm in (("sph", np.zeros(24000, np.int32), np.pad(g["sph_prm"].astype(np.float64), ((0, 0), (0, 1)))),
                           ("rect", (g["rect_axis"] + 1).astype(np.int32), g["rect_prm"])):
        n = kind.shape[0]
        b, pos = kat("shapehits", n, inputs=[kind, prm, g[pre + "_ray_o"], g[pre + "_ray_d"], g[pre + "_tmax"]]), 0
        has, pos = _take(b, pos, np.uint8, (n,))
        t, pos = _take(b, pos, np.float64, (n,))
        p, pos = _take(b, pos, np.float32, (n, 3))
        nrm, pos = _take(b, pos, np.float32, (n, 3))
        ff, pos = _take(b, pos, np.uint8, (n,))
        uv, pos = _take(b, pos, np.float64, (n, 2))
        assert np.array_equal(has, g[pre + "_has"]) and 0.05 < has.mean() < 0.95
        h = has.astype(bool)
        assert O.bits_equal(t[h], g[pre + "_t"][h]) and O.bits_equal(p[h], g[pre + "_p"][h]) and O.bits_equal(nrm[h], g[pre + "_n"][h])
        assert np.array_equal(ff[h], g[pre + "_ff"][h]) and O.bits_equal(uv[h], g[pre + "_uv"][h])


def test_material_and_pdf_classes_match_the_reference(kat):
    """material::scatter / scattering_pdf, onb, cosine_pdf, hittable_pdf over a sphere / XZ / XY (quirk Q2) / YZ (base class) light,
    mixture_pdf, shape::pdf_value / random: the 4000-case call sequence of the compiled reference on engines seeded (21, 22, 23),
    replayed through the mirror headers' host arithmetic."""
    g = O.golden("materials")
    n = g["in_d"].shape[0]
    assert tuple(g["seeds"]) == (21, 22, 23)
    pr = np.ascontiguousarray(g["xzlight_gen_and_o"][:, 1])
    b, pos = kat("materials", n, inputs=[g["in_d"], g["in_n"], g["in_p"], g["in_ff"], g["prm"], g["ir"], g["lamb_sd"], pr]), 0
    for key in ("metal_o", "metal_d", "metal_att", "diel_d", "diel_att", "lamb_att"):
        got, pos = _take(b, pos, np.float32, (n, 3))
        assert O.bits_equal(got, g[key]), key
    for key in ("lamb_spdf", "cos_val"):
        got, pos = _take(b, pos, np.float64, (n,))
        assert O.bits_equal(got, g[key]), key
    got, pos = _take(b, pos, np.float32, (n, 3))
    assert O.bits_equal(got, g["cos_gen"])
    for light in ("sphlight", "xzlight", "xylight", "yzlight"):
        gen, pos = _take(b, pos, np.float32, (n, 3))
        val, pos = _take(b, pos, np.float64, (n,))
        want = g[light + "_gen_and_o"][:, 0] if light == "xzlight" else g[light + "_gen"]
        assert O.bits_equal(gen, want), light
        assert O.same_bits_or_nan(val, g[light + "_val"]), light
    gen, pos = _take(b, pos, np.float32, (n, 3))
    val, pos = _take(b, pos, np.float64, (n,))
    assert O.bits_equal(gen, g["mix_gen"]) and O.same_bits_or_nan(val, g["mix_val"])
    assert (g["sphlight_val"] > 0).mean() > 0.3 and (g["xzlight_val"] > 0).mean() > 0.3
