"""Scene descriptions for the ray path: the reference's scene recipes rebuilt as plain data.

A `Scene` is the Python-side twin of RayView's `m_worldObjects` / `worldLights` / `cam`
(reference rayview.h:62-63,87). It can be
  * flattened to the `RtScene` arrays of include/ray_amd.h (boxes expanded to six rects in the
    constructor order of reference shapes.h:347-354), and
  * written as a text scene file (C99 hex floats, so every value is exact) that
    oracle/ref_harness.cpp turns back into the reference's own objects.

Scene recipes follow reference rayview.cpp: `builtin_scene` = normal_scene (:233-260),
`random_scene` = random_scene (:186-231) with the grid range as a parameter, `box_scene` =
box_scene (:262-378). `tri_mesh_scene` / `mixed_scene` are new (the reference has no triangles).
All float32 arithmetic of the recipes is done in numpy float32 so values are reproducible.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32

PRIM_SPHERE, PRIM_XY_RECT, PRIM_XZ_RECT, PRIM_YZ_RECT, PRIM_TRIANGLE, PRIM_MEDIUM_SPHERE, PRIM_MEDIUM_BOX = 0, 1, 2, 3, 4, 5, 6
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
LIGHT_BASE = -1
PRIM_STRIDE = 9

_KIND_NAMES = {PRIM_SPHERE: "sphere", PRIM_XY_RECT: "xy_rect", PRIM_XZ_RECT: "xz_rect",
               PRIM_YZ_RECT: "yz_rect", PRIM_TRIANGLE: "triangle"}
_MAT_NAMES = {MAT_LAMBERTIAN: "lambertian", MAT_METAL: "metal", MAT_DIELECTRIC: "dielectric",
              MAT_DIFFUSE_LIGHT: "diffuse_light", MAT_ISOTROPIC: "isotropic"}


# ------------------------------------------------------------------------------------------------
# counter RNG (the same generator the HIP kernels and the oracle use; DESIGN.md "RNG contract")
# ------------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def pcg_hash(v: int) -> int:
    state = (v * 747796405 + 2891336453) & _M32
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & _M32
    return ((word >> 22) ^ word) & _M32


class CounterRng:
    """Stream keyed by (seed, pixel, sample); next() is PCG-RXS-M-XS-32."""

    def __init__(self, seed: int, pixel: int = 0, sample: int = 0):
        lo, hi = seed & _M32, (seed >> 32) & _M32
        self.state = pcg_hash((pixel + pcg_hash((sample + pcg_hash(lo ^ pcg_hash(hi))) & _M32)) & _M32)

    def bits(self) -> int:
        self.state = (self.state * 747796405 + 2891336453) & _M32
        s = self.state
        word = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & _M32
        return ((word >> 22) ^ word) & _M32

    def u01(self) -> np.float32:
        return f32(self.bits() >> 8) * f32(2.0 ** -24)

    def uniform(self, a, b) -> np.float32:
        a, b = f32(a), f32(b)
        return f32(f32(self.u01() * f32(b - a)) + a)


# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class Material:
    kind: int
    rgb: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    fuzz: float = 0.0          # as passed to metal(a, f); clamped like hitPosition.h:226 on flatten
    ir: float = 1.0


@dataclasses.dataclass
class Shape:
    kind: str                  # sphere | xy_rect | xz_rect | yz_rect | box | triangle | medium_sphere | medium_box
    data: Tuple[float, ...]    # sphere: cx cy cz r; rect: a0 a1 b0 b1 k; box: p0xyz p1xyz; tri: 9;
                               # medium_sphere: cx cy cz r density; medium_box: p0xyz p1xyz density (constant_medium,
                               # shapes.h:374; mat = an isotropic material holding the colour)
    mat: int = -1


@dataclasses.dataclass
class Camera:
    lookfrom: Tuple[float, float, float]
    lookat: Tuple[float, float, float]
    vup: Tuple[float, float, float]
    focus: float
    vfov: float = 40.0
    aspect: float = float(f32(16.0) / f32(9.0))   # img::aspect_ratio, calc.h:15
    aperture: float = 0.1


@dataclasses.dataclass
class Scene:
    mats: List[Material]
    world: List[Shape]
    light: Optional[Shape]
    camera: Camera
    background: Tuple[float, float, float] = (0.5, 0.7, 1.0)   # img::gradientBgVec, calc.h:22
    name: str = "scene"

    # -- text form for oracle/ref_harness.cpp ---------------------------------------------------
    def to_text(self) -> str:
        def h(x):
            return float(x).hex()

        def h32(x):
            return float(f32(x)).hex()

        out = ["rtscene 1"]
        for m in self.mats:
            if m.kind == MAT_ISOTROPIC:
                out.append("lambertian " + " ".join(h32(c) for c in m.rgb))   # placeholder slot: a medium carries its own colour
            elif m.kind == MAT_DIELECTRIC:
                out.append(f"dielectric {h(m.ir)}")
            elif m.kind == MAT_METAL:
                out.append("metal " + " ".join(h32(c) for c in m.rgb) + f" {h(m.fuzz)}")
            else:
                out.append(_MAT_NAMES[m.kind] + " " + " ".join(h32(c) for c in m.rgb))

        def shape_line(s: Shape, with_mat: bool) -> str:
            if s.kind == "triangle":
                raise ValueError("the reference has no triangle primitive")
            if s.kind in ("medium_sphere", "medium_box"):   # constant_medium(boundary, density, colour)
                rgb = self.mats[s.mat].rgb
                return f"{s.kind} " + " ".join(h32(v) for v in s.data[:-1]) + f" {h(s.data[-1])} " + " ".join(h32(c) for c in rgb)
            if s.kind in ("sphere", "box"):
                vals = " ".join(h32(v) for v in s.data)
            else:
                vals = " ".join(h(v) for v in s.data)
            return f"{s.kind} {vals}" + (f" {s.mat}" if with_mat else "")

        if self.light is not None:
            out.append("light " + shape_line(self.light, False))
        c = self.camera
        out.append("camera " + " ".join(h32(v) for v in (*c.lookfrom, *c.lookat, *c.vup))
                   + f" {h(c.focus)} {h(c.vfov)} {h(c.aspect)} {h(c.aperture)}")
        out.append("background " + " ".join(h32(v) for v in self.background))
        for s in self.world:
            out.append(shape_line(s, True))
        return "\n".join(out) + "\n"

    # -- flattened arrays of include/ray_amd.h ---------------------------------------------------
    def flatten(self) -> dict:
        kind, mat, outer, side, data = [], [], [], [], []

        def add(k, m, o, sd, vals):
            row = [0.0] * PRIM_STRIDE
            row[:len(vals)] = [float(v) for v in vals]
            kind.append(k); mat.append(m); outer.append(o); side.append(sd); data.append(row)

        for i, s in enumerate(self.world):
            if s.kind == "sphere":
                add(PRIM_SPHERE, s.mat, i, -1, [f32(v) for v in s.data])
            elif s.kind == "xy_rect":
                add(PRIM_XY_RECT, s.mat, i, -1, s.data)
            elif s.kind == "xz_rect":
                add(PRIM_XZ_RECT, s.mat, i, -1, s.data)
            elif s.kind == "yz_rect":
                add(PRIM_YZ_RECT, s.mat, i, -1, s.data)
            elif s.kind == "triangle":
                add(PRIM_TRIANGLE, s.mat, i, -1, [f32(v) for v in s.data])
            elif s.kind in ("medium_sphere", "medium_box"):
                if self.mats[s.mat].kind != MAT_ISOTROPIC:
                    raise ValueError("a constant_medium's material is its isotropic phase function")
                geo = [float(f32(v)) for v in s.data[:-1]]
                add(PRIM_MEDIUM_SPHERE if s.kind == "medium_sphere" else PRIM_MEDIUM_BOX, s.mat, i, -1, geo + [-1.0 / float(s.data[-1])])
            elif s.kind == "box":
                # reference shapes.h:347-354: constructor order of the six sides
                p0 = [float(f32(v)) for v in s.data[0:3]]
                p1 = [float(f32(v)) for v in s.data[3:6]]
                add(PRIM_XY_RECT, s.mat, i, 0, [p0[0], p1[0], p0[1], p1[1], p1[2]])
                add(PRIM_XY_RECT, s.mat, i, 1, [p0[0], p1[0], p0[1], p1[1], p0[2]])
                add(PRIM_XZ_RECT, s.mat, i, 2, [p0[0], p1[0], p0[2], p1[2], p1[1]])
                add(PRIM_XZ_RECT, s.mat, i, 3, [p0[0], p1[0], p0[2], p1[2], p0[1]])
                add(PRIM_YZ_RECT, s.mat, i, 4, [p0[1], p1[1], p0[2], p1[2], p1[0]])
                add(PRIM_YZ_RECT, s.mat, i, 5, [p0[1], p1[1], p0[2], p1[2], p0[0]])
            else:
                raise ValueError(f"unknown shape kind {s.kind}")
        n = len(kind)
        light_kind, light_data = LIGHT_BASE, [0.0] * PRIM_STRIDE
        if self.light is not None:
            lk = {"sphere": PRIM_SPHERE, "xy_rect": PRIM_XY_RECT, "xz_rect": PRIM_XZ_RECT}.get(self.light.kind, LIGHT_BASE)
            light_kind = lk
            vals = [f32(v) for v in self.light.data] if self.light.kind == "sphere" else list(self.light.data)
            if lk != LIGHT_BASE:
                light_data[:len(vals)] = [float(v) for v in vals]
        mk = np.array([m.kind for m in self.mats], dtype=np.int32)
        mrgb = np.array([[f32(c) for c in m.rgb] for m in self.mats], dtype=np.float32).reshape(-1, 3)
        # metal(a, f): fuzz(f < 1.0f ? f : 1.0f) stored as float (hitPosition.h:224-227)
        mf = np.array([f32(m.fuzz) if m.fuzz < 1.0 else f32(1.0) for m in self.mats], dtype=np.float32)
        mir = np.array([m.ir for m in self.mats], dtype=np.float64)
        return dict(
            n_prims=n,
            prim_kind=np.array(kind, dtype=np.int32), prim_mat=np.array(mat, dtype=np.int32),
            prim_outer=np.array(outer, dtype=np.int32), prim_side=np.array(side, dtype=np.int32),
            prim_data=np.array(data, dtype=np.float64).reshape(n, PRIM_STRIDE),
            n_mats=len(self.mats), mat_kind=mk, mat_rgb=mrgb, mat_fuzz=mf, mat_ir=mir,
            light_kind=light_kind, light_data=np.array(light_data, dtype=np.float64),
            background=np.array([f32(c) for c in self.background], dtype=np.float32),
        )


# ------------------------------------------------------------------------------------------------
# recipes
# ------------------------------------------------------------------------------------------------
LIGHT_TARGET = Shape("sphere", (190.0, 90.0, 190.0, 90.0))      # rayview.cpp:49


def rayview_camera(aperture: float = 0.1) -> Camera:
    # rayview.cpp:18-20,27: focus = (lookFrom - lookAt).length() = 2
    return Camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 2.0, 40.0,
                  float(f32(16.0) / f32(9.0)), aperture)


def builtin_scene(sphere_y: float = 0.5, aperture: float = 0.1, light: str = "target") -> Scene:
    """normal_scene(), rayview.cpp:233-260, in push_back order. `light` selects the shape passed as
    calc::ray_color's `lights`: "target" = the sphere of rayview.cpp:49; "xy" = the scene's own emitting
    `rectangle` (exercises rectangle::pdf_value / random, shapes.h:202-219, incl. the k-in-the-y-slot quirk);
    "yz" = a yz_rect, which has neither (base class shape: value 0, direction (1,0,0), shapes.h:56-62)."""
    mats = [
        Material(MAT_LAMBERTIAN, (0.8, 0.8, 0.0)),      # 0 ground
        Material(MAT_LAMBERTIAN, (0.7, 0.3, 0.3)),      # 1 center
        Material(MAT_DIELECTRIC, ir=1.5),               # 2 left
        Material(MAT_METAL, (0.6, 0.6, 0.6), fuzz=0.0),  # 3 right
        Material(MAT_DIFFUSE_LIGHT, (2.0, 2.0, 2.0)),   # 4 light
    ]
    world = [
        Shape("sphere", (0.0, sphere_y, -1.0, 0.5), 1),
        Shape("sphere", (0.0, -100.2, -1.0, 100.0), 0),
        Shape("sphere", (1.0, 0.0, -1.0, 0.6), 3),
        Shape("sphere", (-1.0, 0.0, -1.0, 0.6), 2),
        Shape("xy_rect", (3.0, 5.0, 1.0, 3.0, -2.0), 4),
    ]
    lights = {"target": LIGHT_TARGET, "xy": Shape("xy_rect", (3.0, 5.0, 1.0, 3.0, -2.0)),
              "yz": Shape("yz_rect", (1.0, 3.0, -3.0, -1.0, 4.0))}
    return Scene(mats, world, lights[light], rayview_camera(aperture), name="builtin" if light == "target" else f"builtin_{light}light")


def random_scene(half: int = 11, seed: int = 1, aperture: float = 0.1) -> Scene:
    """random_scene(), rayview.cpp:186-231, with the grid range [-half, half) (the reference
    uses 5; half=11 gives the ~485-sphere benchmark scene of SURVEY.md section 8d). The scene's
    random numbers come from CounterRng(seed) instead of the reference's unseeded engines."""
    rng = CounterRng(seed)
    mats = [Material(MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]
    world = [Shape("sphere", (0.0, -1000.0, 0.0, 1000.0), 0)]
    ref = np.array([4.0, 0.2, 0.0], dtype=np.float32)
    for a in range(-half, half):
        for b in range(-half, half):
            choose = rng.u01()
            cx = f32(f32(a) + f32(0.9) * rng.u01())
            cz = f32(f32(b) + f32(0.9) * rng.u01())
            c = np.array([cx, f32(0.2), cz], dtype=np.float32)
            d = (c - ref).astype(np.float64)
            if f32(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) > f32(0.9):
                if choose < f32(0.8):
                    a1 = [rng.u01() for _ in range(3)]
                    a2 = [rng.u01() for _ in range(3)]
                    alb = tuple(float(f32(x * y)) for x, y in zip(a1, a2))
                    mats.append(Material(MAT_LAMBERTIAN, alb))
                elif choose < f32(0.95):
                    alb = tuple(float(rng.uniform(0.5, 1.0)) for _ in range(3))
                    fuzz = float(rng.uniform(0.0, 0.5))
                    mats.append(Material(MAT_METAL, alb, fuzz=fuzz))
                else:
                    mats.append(Material(MAT_DIELECTRIC, ir=1.5))
                world.append(Shape("sphere", (float(cx), 0.2, float(cz), 0.2), len(mats) - 1))
    mats.append(Material(MAT_DIELECTRIC, ir=1.5))
    world.append(Shape("sphere", (0.0, 1.0, 0.0, 1.0), len(mats) - 1))
    mats.append(Material(MAT_LAMBERTIAN, (0.4, 0.2, 0.1)))
    world.append(Shape("sphere", (-4.0, 1.0, 0.0, 1.0), len(mats) - 1))
    mats.append(Material(MAT_METAL, (0.7, 0.6, 0.5), fuzz=0.0))
    world.append(Shape("sphere", (4.0, 1.0, 0.0, 1.0), len(mats) - 1))
    cam = Camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 10.0, 20.0, 16.0 / 9.0, aperture)
    return Scene(mats, world, LIGHT_TARGET, cam, name=f"random{half}")


def box_scene(aperture: float = 0.0, light_on_panel: bool = False) -> Scene:
    """box_scene(), rayview.cpp:262-378 ("CPP NORTH"), 13 rects/spheres + 26 boxes.
    light_on_panel=True puts the sampled light rect exactly on a ceiling panel of the world, so
    light directions from that panel lie in its plane (t = 0/0): the reference's NaN-accepting
    comparisons (shapes.h:241,246) then report hits with t = NaN -- an edge case kept as a golden."""
    mats = [
        Material(MAT_LAMBERTIAN, (.65, .05, .05)),      # 0 red
        Material(MAT_LAMBERTIAN, (.73, .73, .73)),      # 1 white
        Material(MAT_LAMBERTIAN, (.12, .45, .15)),      # 2 green
        Material(MAT_LAMBERTIAN, (.77, .77, .77)),      # 3 gray
        Material(MAT_DIFFUSE_LIGHT, (20., 20., 20.)),   # 4 light
        Material(MAT_METAL, (0.6, 0.6, 0.6), fuzz=0.0),  # 5 metal
    ]
    RED, WHITE, GREEN, GRAY, LIGHT, METAL = range(6)
    w: List[Shape] = [
        Shape("yz_rect", (0, 600, -600, 3000, 600), RED),
        Shape("yz_rect", (0, 600, -600, 3000, 0), GREEN),
        Shape("sphere", (275, 3000, 1000, 500), LIGHT),
        Shape("xz_rect", (0, 600, -600, 3000, 0), METAL),
        Shape("xz_rect", (0, 200, -600, 3000, 600), WHITE),
        Shape("xz_rect", (400, 600, -600, 3000, 600), WHITE),
        Shape("xz_rect", (200, 400, -600, -200, 580), GRAY),
        Shape("xz_rect", (200, 400, 200, 400, 580), GRAY),
        Shape("xz_rect", (200, 400, 1000, 1200, 580), GRAY),
        Shape("xz_rect", (200, 400, 1800, 2000, 580), GRAY),
        Shape("xz_rect", (200, 400, 2600, 2800, 580), GRAY),
        Shape("xy_rect", (-600, 600, 0, 600, 3000), WHITE),
        Shape("xy_rect", (-600, 600, 0, 600, -1000), LIGHT),
    ]
    sf = f32(0.22)

    def v(x, y, z):
        return np.array([x, y, z], dtype=np.float32)

    pos = v(0, 250, 700)

    def bx(a, b, m):
        p0 = (pos + a * sf).astype(np.float32) if a is not None else pos.copy()
        p1 = (pos + b * sf).astype(np.float32)
        w.append(Shape("box", tuple(float(t) for t in (*p0, *p1)), m))

    # C
    bx(None, v(100, 200, 100), RED); bx(v(100, 0, 0), v(200, 50, 100), RED); bx(v(100, 150, 0), v(200, 200, 100), RED)
    pos = (pos + v(300, 0, 0) * sf).astype(np.float32)
    for _ in range(2):   # P P
        bx(v(0, 0, 0), v(100, 200, 100), RED); bx(v(100, 150, 0), v(200, 200, 100), RED)
        bx(v(100, 50, 0), v(200, 100, 100), RED); bx(v(150, 100, 0), v(200, 150, 100), RED)
        pos = (pos + v(300 if _ == 0 else 350, 0, 0) * sf).astype(np.float32)
    # N
    bx(v(0, 0, 0), v(100, 200, 100), GREEN); bx(v(100, 125, 0), v(250, 200, 100), GREEN); bx(v(150, 0, 0), v(250, 150, 100), GREEN)
    pos = (pos + v(350, 0, 0) * sf).astype(np.float32)
    # O
    bx(v(0, 0, 0), v(100, 200, 100), GREEN); bx(v(100, 150, 0), v(300, 200, 100), GREEN)
    bx(v(200, 0, 0), v(300, 200, 100), GREEN); bx(v(100, 0, 0), v(300, 50, 100), GREEN)
    pos = (pos + v(400, 0, 0) * sf).astype(np.float32)
    # R
    bx(v(0, 0, 0), v(100, 200, 100), GREEN); bx(v(100, 150, 0), v(200, 200, 100), GREEN); bx(v(150, 100, 0), v(200, 150, 100), GREEN)
    pos = (pos + v(300, 0, 0) * sf).astype(np.float32)
    # T
    bx(v(75, 0, 0), v(125, 200, 100), GREEN); bx(v(0, 150, 0), v(200, 200, 100), GREEN)
    pos = (pos + v(300, 0, 0) * sf).astype(np.float32)
    # H
    bx(v(0, 0, 0), v(100, 200, 100), GREEN); bx(v(200, 0, 0), v(300, 200, 100), GREEN); bx(v(100, 75, 0), v(200, 125, 100), GREEN)
    # the reference fixes no camera for this scene; this one looks down the corridor at the letters
    cam = Camera((300.0, 300.0, -500.0), (300.0, 280.0, 700.0), (0.0, 1.0, 0.0), 1200.0, 40.0, 16.0 / 9.0, aperture)
    light = Shape("xz_rect", (200, 400, 200, 400, 580) if light_on_panel else (213, 343, 227, 332, 554))
    return Scene(mats, w, light, cam, name="box")


def smoke_scene(aperture: float = 0.0, density: float = 0.6) -> Scene:
    """A small room in the manner of box_scene with the reference's remaining shape / material pair in it
    (SURVEY.md 8f row 3): a box of dark smoke and a ball of white fog (constant_medium over a box and over a sphere,
    shapes.h:374-455, isotropic phase function hitPosition.h:316-337), a light panel, a metal ball, a glass ball
    inside the fog's reach. The reference ships no such scene; this one is the build's."""
    mats = [
        Material(MAT_LAMBERTIAN, (.65, .05, .05)), Material(MAT_LAMBERTIAN, (.73, .73, .73)), Material(MAT_LAMBERTIAN, (.12, .45, .15)),
        Material(MAT_DIFFUSE_LIGHT, (7., 7., 7.)), Material(MAT_METAL, (0.8, 0.85, 0.88), fuzz=0.1), Material(MAT_DIELECTRIC, ir=1.5),
        Material(MAT_ISOTROPIC, (0.05, 0.05, 0.05)), Material(MAT_ISOTROPIC, (1.0, 1.0, 1.0)),
    ]
    RED, WHITE, GREEN, LIGHT, METAL, GLASS, SMOKE, FOG = range(8)
    w = [
        Shape("yz_rect", (0, 5.55, 0, 5.55, 5.55), GREEN), Shape("yz_rect", (0, 5.55, 0, 5.55, 0), RED),
        Shape("xz_rect", (1.13, 4.43, 1.27, 4.32, 5.54), LIGHT),
        Shape("xz_rect", (0, 5.55, 0, 5.55, 0), WHITE), Shape("xz_rect", (0, 5.55, 0, 5.55, 5.55), WHITE),
        Shape("xy_rect", (0, 5.55, 0, 5.55, 5.55), WHITE),
        Shape("medium_box", (2.65, 0.0, 2.95, 4.30, 3.30, 4.60, density), SMOKE),
        Shape("sphere", (1.6, 0.9, 1.9, 0.9), GLASS),
        Shape("medium_sphere", (1.6, 0.9, 1.9, 1.4, 0.35 * density), FOG),      # fog around (and inside) the glass ball
        Shape("sphere", (3.9, 0.6, 1.4, 0.6), METAL),
        Shape("medium_sphere", (2.8, 2.8, 2.8, 30.0, 0.004 * density), FOG),    # thin haze that the camera sits inside
    ]
    cam = Camera((2.78, 2.78, -8.0), (2.78, 2.78, 0.0), (0.0, 1.0, 0.0), 10.0, 40.0, 16.0 / 9.0, aperture)
    return Scene(mats, w, Shape("xz_rect", (1.13, 4.43, 1.27, 4.32, 5.54)), cam, (0.0, 0.0, 0.0), name="smoke")


def _height(x: np.ndarray, z: np.ndarray) -> np.ndarray:
    return (0.35 * np.sin(1.3 * x) * np.cos(0.9 * z) + 0.2 * np.sin(2.1 * x + 0.7 * z) + 0.6).astype(np.float32)


def tri_mesh(n: int = 71, extent: float = 5.0, mat: int = 0) -> List[Shape]:
    """n x n height-field grid -> 2 (n-1)^2 triangles (n=71: 9800)."""
    xs = np.linspace(-extent, extent, n).astype(np.float32)
    zs = np.linspace(-extent, extent, n).astype(np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = _height(X.astype(np.float64), Z.astype(np.float64))
    tris = []
    for i in range(n - 1):
        for j in range(n - 1):
            p00 = (X[i, j], Y[i, j], Z[i, j]); p10 = (X[i + 1, j], Y[i + 1, j], Z[i + 1, j])
            p01 = (X[i, j + 1], Y[i, j + 1], Z[i, j + 1]); p11 = (X[i + 1, j + 1], Y[i + 1, j + 1], Z[i + 1, j + 1])
            tris.append(Shape("triangle", tuple(float(t) for t in (*p00, *p01, *p10)), mat))
            tris.append(Shape("triangle", tuple(float(t) for t in (*p10, *p01, *p11)), mat))
    return tris


def tri_mesh_scene(n: int = 71, aperture: float = 0.1) -> Scene:
    """S-tris (BASELINE.json configs[3]): ~10k-triangle height field over a ground sphere."""
    mats = [Material(MAT_LAMBERTIAN, (0.5, 0.5, 0.5)), Material(MAT_LAMBERTIAN, (0.2, 0.5, 0.7)),
            Material(MAT_METAL, (0.8, 0.8, 0.8), fuzz=0.05)]
    world = [Shape("sphere", (0.0, -1000.0, 0.0, 1000.0), 0)]
    world += tri_mesh(n, 5.0, 1)
    world.append(Shape("sphere", (0.0, 2.0, 0.0, 0.8), 2))
    cam = Camera((9.0, 5.0, 9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 13.0, 35.0, 16.0 / 9.0, aperture)
    return Scene(mats, world, LIGHT_TARGET, cam, name=f"tris{n}")


def mixed_scene(half: int = 11, n: int = 71, seed: int = 1, aperture: float = 0.1) -> Scene:
    """S-mixed (BASELINE.json configs[4]): the random spheres plus the triangle mesh, raised."""
    sc = random_scene(half, seed, aperture)
    mats = list(sc.mats) + [Material(MAT_LAMBERTIAN, (0.2, 0.5, 0.7))]
    tris = tri_mesh(n, 3.0, len(mats) - 1)
    lifted = []
    for t in tris:
        d = list(t.data)
        for k in (1, 4, 7):
            d[k] = float(f32(d[k]) + f32(2.2))
        lifted.append(Shape("triangle", tuple(d), t.mat))
    return Scene(mats, list(sc.world) + lifted, sc.light, sc.camera, sc.background, name=f"mixed{half}_{n}")


def get_scene(name: str, **kw) -> Scene:
    table = {"builtin": builtin_scene, "random": random_scene, "box": box_scene,
             "tris": tri_mesh_scene, "mixed": mixed_scene, "smoke": smoke_scene}
    return table[name](**kw)
