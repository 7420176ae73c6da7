#!/usr/bin/env python3
"""Differential campaign: random scenes (the repository's generators with random parameters, plus random soups of every shape and
material kind, random cameras and lights), small frames, GPU accumulators against the CPU oracle bit for bit (NaN where NaN).
    python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ray_amd import api, scenes  # noqa: E402
import oracle_lib as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ctx = api.Context(0)


def f32(x):
    return float(np.float32(x))


def soup(rng):
    mats = [scenes.Material(scenes.MAT_LAMBERTIAN, tuple(rng.uniform(0.1, 0.9, 3))),
            scenes.Material(scenes.MAT_METAL, tuple(rng.uniform(0.5, 1.0, 3)), f32(rng.uniform(0, 0.6))),
            scenes.Material(scenes.MAT_DIELECTRIC, (1.0, 1.0, 1.0), 0.0, f32(rng.uniform(1.1, 2.0))),
            scenes.Material(scenes.MAT_DIFFUSE_LIGHT, tuple(rng.uniform(1.0, 6.0, 3)))]
    shapes = [scenes.Shape("sphere", (0.0, -1000.5, -1.0, 1000.0), 0)]
    for _ in range(int(rng.integers(1, 60))):
        c = rng.uniform(-4, 4, 3); c[1] = abs(c[1]) * 0.5; c[2] -= 3.0
        shapes.append(scenes.Shape("sphere", (f32(c[0]), f32(c[1]), f32(c[2]), f32(rng.uniform(0.1, 0.9))), int(rng.integers(0, 3))))
    for _ in range(int(rng.integers(0, 30))):
        kind = ["xy_rect", "xz_rect", "yz_rect"][int(rng.integers(0, 3))]
        a0, b0 = rng.uniform(-4, 3, 2)
        shapes.append(scenes.Shape(kind, (f32(a0), f32(a0 + rng.uniform(0.2, 2.5)), f32(b0), f32(b0 + rng.uniform(0.2, 2.5)), f32(rng.uniform(-5, 1))),
                                   int(rng.integers(0, 4))))
    for _ in range(int(rng.integers(0, 80))):
        p = rng.uniform(-3, 3, 3); p[2] -= 3.0
        e1, e2 = rng.standard_normal(3) * 0.7, rng.standard_normal(3) * 0.7
        v = [np.float32(p), np.float32(p + e1), np.float32(p + e2)]
        shapes.append(scenes.Shape("triangle", tuple(float(t) for t in (*v[0], *v[1], *v[2])), int(rng.integers(0, 3))))
    order = rng.permutation(len(shapes))
    shapes = [shapes[i] for i in order]
    light_kind = int(rng.integers(0, 3))
    if light_kind == 0:
        light = scenes.Shape("sphere", (f32(rng.uniform(-3, 3)), f32(rng.uniform(3, 8)), f32(rng.uniform(-4, 1)), f32(rng.uniform(0.5, 2.0))), 3)
    elif light_kind == 1:
        light = scenes.Shape("xz_rect", (-1.0, 1.5, -4.0, -2.0, f32(rng.uniform(3, 6))), 3)
        shapes.append(light)
    else:
        light = scenes.Shape("xy_rect", (2.0, 4.0, 0.5, 2.5, -4.5), 3)
        shapes.append(light)
    cam = scenes.rayview_camera(aperture=f32(rng.choice([0.0, 0.05, 0.3])))
    return scenes.Scene(mats, shapes, light, cam)


def case(i):
    k = int(rng.integers(0, 7))
    if k == 0:
        return "random", scenes.random_scene(half=int(rng.integers(1, 7)), seed=int(rng.integers(1, 1000)), aperture=f32(rng.choice([0.0, 0.1])))
    if k == 1:
        return "builtin", scenes.builtin_scene(sphere_y=f32(rng.uniform(-0.5, 1.5)), light=str(rng.choice(["target", "xy", "yz"])))
    if k == 2:
        return "box", scenes.box_scene(light_on_panel=bool(rng.integers(0, 2)))
    if k == 3:
        if os.environ.get("RT_FUZZ_MESHES") == "1":   # meshes of several filter chunks (576 .. 5000 items): the shared level 0 of the triangle filter and its decision per scan
            if rng.integers(0, 2):
                return "tris", scenes.tri_mesh_scene(n=int(rng.integers(24, 72)), aperture=f32(rng.choice([0.0, 0.1])))
            return "mixed", scenes.mixed_scene(half=int(rng.integers(1, 5)), n=int(rng.integers(24, 72)), seed=int(rng.integers(1, 1000)))
        return "mixed", scenes.mixed_scene(half=int(rng.integers(1, 5)), n=int(rng.integers(3, 30)), seed=int(rng.integers(1, 1000)))
    if k == 4:
        return "smoke", scenes.smoke_scene(density=f32(rng.uniform(0.05, 2.0)))
    return "soup", soup(rng)


bad = 0
t0 = time.time()
total_rays = 0
for i in range(n_cases):
    name, sc = case(i)
    W, H = int(rng.integers(8, 97)), int(rng.integers(5, 61))
    spp, depth = int(rng.integers(1, 5)), int(rng.choice([1, 2, 5, 8, 12, 20]))
    how = ""
    if os.environ.get("RT_FUZZ_UNITS") == "1":   # round 3: how the frame's work is cut up -- colour buffer of the frame's end, sample slices, shading sort
        spp = int(rng.choice([1, 3, 8, 16, 17, 24, 40]))
        mb, smin, smax, sort = int(rng.choice([0, 1, 2048])), int(rng.choice([0, 1, 3, 8])), int(rng.choice([2, 5, 16])), int(rng.integers(0, 2))
        order = int(rng.choice([0, 1, 2, 2]))
        ctx.set_tail_limit(mb); ctx.set_sample_slices(smin, smax); ctx.set_shading_sort(sort); ctx.set_tile_order(order)
        how = f" tail {mb} MiB slices ({smin}, {smax}) sort {sort} order {order}"
    seed = int(rng.integers(0, 2 ** 62))
    ctx.upload_scene(sc)
    acc, rays = ctx.render(W, H, spp, depth, seed=seed)
    oacc, orays = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=seed, math_mode=1)
    ok = rays == orays and O.same_bits_or_nan(acc, oacc)
    total_rays += rays
    what = ""
    if ok and i % 4 == 0:       # the other entry points on the same scene: primary ids / t, closest hits of random rays, radiance of random rays, a pixel list
        prim, t = ctx.primary_hits(W, H, seed=seed)
        oprim, ot = O.primary(sc, W, H, rng_mode=1, seed=seed)
        if not (np.array_equal(prim, oprim) and O.bits_equal(t, ot)):
            ok, what = False, "primary_hits"
        n = 600
        ro = rng.uniform(-6, 6, (n, 3)).astype(np.float32); ro[:, 2] -= 1.0
        rd = rng.standard_normal((n, 3)).astype(np.float32)
        rd[::17, int(rng.integers(0, 3))] = 0.0
        t_min, t_max = [(0.001, np.inf), (-3.0, 25.0), (0.5, 7.0)][int(rng.integers(0, 3))]
        got = ctx.hit_rays(ro, rd, t_min, t_max)
        ref = O.hit_rays(sc, ro, rd, t_min, t_max)
        if not (np.array_equal(got[0], ref[0]) and O.bits_equal(got[1], ref[1]) and all(O.same_bits_or_nan(a, b) for a, b in zip(got[2:4], ref[2:4]))
                and np.array_equal(got[4], ref[4])):
            ok, what = False, what + " hit_rays"
        g = ctx.trace_rays(ro, rd, depth, seed=seed & 0xffffffff)
        r = O.trace_rays(sc, ro, rd, depth, seed=seed & 0xffffffff)
        if not O.same_bits_or_nan(g, r):
            ok, what = False, what + " trace_rays"
        px = rng.integers(0, W * H, int(rng.integers(1, 4000))).astype(np.int32)
        a2 = ctx.alloc(W * H * 16).zero()
        ctx.render_pixels(W, H, a2, px, spp, depth, seed=seed)
        g2 = a2.to_numpy(np.float32, (H, W, 4)).reshape(-1, 4); a2.free()
        m = np.zeros(W * H, bool); m[px] = True
        if not (O.same_bits_or_nan(g2[m], acc.reshape(-1, 4)[m]) and (g2[~m] == 0).all()):
            ok, what = False, what + " render_pixels"
    if not ok:
        bad += 1
        diff = int((acc.view(np.uint32) != oacc.view(np.uint32)).sum())
        print("   failing:", what or "render")
    print(f"case {i:3d} {name:8s} prims {sc.flatten()['n_prims']:5d} {W}x{H} spp {spp} depth {depth:2d}{how} rays {rays:8d} {'ok' if ok else 'MISMATCH (%d words, rays %d vs %d)' % (diff, rays, orays)}", flush=True)
print(f"{n_cases} cases, {total_rays} rays, {bad} mismatching frames, {time.time() - t0:.0f} s")
ctx.close()
sys.exit(1 if bad else 0)
