"""Analytic known answers for the triangle primitive (not in the reference; include/ray_amd_tri.h), shared by
the oracle test (CPU) and the HIP test (GPU). Nothing here goes through either implementation: the expected
values are hand-computed on geometry whose FP32 arithmetic is exact, an independent float64 ray-plane /
barycentric computation, and the reference's own `rectangle` (shapes.h:172-192) as a second opinion.

`hit_rays(scene, ro, rd, t_min, t_max)` -> (prim, t, p, normal, front_face) is the implementation under test."""
import numpy as np

from ray_amd import scenes

_MATS = [scenes.Material(scenes.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))]


def _scene(shapes):
    return scenes.Scene(_MATS, shapes, None, scenes.rayview_camera())


UNIT = scenes.Shape("triangle", (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0), 0)   # v0, v1, v2 in the plane z = 0, normal +z
INF = float("inf")

# (origin, direction, t_min, t_max, hit?, t, p, normal, front_face) -- every value exact in FP32
UNIT_CASES = [
    ((0.25, 0.25, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (0.25, 0.25, 0.0), (0, 0, 1), True),      # interior
    ((0.25, 0.25, 2.0), (0, 0, -0.5), 0.001, INF, True, 4.0, (0.25, 0.25, 0.0), (0, 0, 1), True),    # unnormalised direction
    ((0.25, 0.25, -1.0), (0, 0, 1), 0.001, INF, True, 1.0, (0.25, 0.25, 0.0), (0, 0, -1), False),    # back face
    ((0.5, 0.0, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (0.5, 0.0, 0.0), (0, 0, 1), True),          # edge v = 0
    ((0.0, 0.5, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (0.0, 0.5, 0.0), (0, 0, 1), True),          # edge u = 0
    ((0.5, 0.5, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (0.5, 0.5, 0.0), (0, 0, 1), True),          # edge u + v = 1
    ((0.0, 0.0, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (0.0, 0.0, 0.0), (0, 0, 1), True),          # vertex v0
    ((1.0, 0.0, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (1.0, 0.0, 0.0), (0, 0, 1), True),          # vertex v1
    ((0.0, 1.0, 1.0), (0, 0, -1), 0.001, INF, True, 1.0, (0.0, 1.0, 0.0), (0, 0, 1), True),          # vertex v2
    ((0.25, 0.25, 1.0), (0.25, 0.25, -1), 0.001, INF, True, 1.0, (0.5, 0.5, 0.0), (0, 0, 1), True),  # oblique, lands on the hypotenuse
    ((-0.0078125, 0.25, 1.0), (0, 0, -1), 0.001, INF, False, 0, None, None, None),                   # u < 0
    ((0.25, -0.0078125, 1.0), (0, 0, -1), 0.001, INF, False, 0, None, None, None),                   # v < 0
    ((0.5078125, 0.5, 1.0), (0, 0, -1), 0.001, INF, False, 0, None, None, None),                     # u + v > 1
    ((-1.0, 0.25, 0.5), (1, 0, 0), 0.001, INF, False, 0, None, None, None),                          # parallel to the plane
    ((-1.0, 0.25, 0.0), (1, 0, 0), 0.001, INF, False, 0, None, None, None),                          # IN the plane (det = 0)
    ((0.25, 0.25, 1.0), (0, 0, -1), 0.001, 1.0, True, 1.0, (0.25, 0.25, 0.0), (0, 0, 1), True),      # t == t_max is inside
    ((0.25, 0.25, 1.0), (0, 0, -1), 0.001, 0.999, False, 0, None, None, None),
    ((0.25, 0.25, 1.0), (0, 0, -1), 1.0, INF, True, 1.0, (0.25, 0.25, 0.0), (0, 0, 1), True),        # t == t_min is inside
    ((0.25, 0.25, 1.0), (0, 0, -1), 1.001, INF, False, 0, None, None, None),
    ((0.25, 0.25, -1.0), (0, 0, -1), 0.001, INF, False, 0, None, None, None),                        # behind the origin
    ((0.25, 0.25, -1.0), (0, 0, -1), -5.0, INF, True, -1.0, (0.25, 0.25, 0.0), (0, 0, 1), True),     # ... unless t_min allows it
    ((0.25, 0.25, 1.0), (0, 0, 0), 0.001, INF, False, 0, None, None, None),                          # zero direction
]


def check_unit_triangle(hit_rays):
    sc = _scene([UNIT])
    for o, d, t0, t1, hit, t, p, n, ff in UNIT_CASES:
        ro, rd = np.array([o], np.float32), np.array([d], np.float32)
        prim, gt, gp, gn, gff = hit_rays(sc, ro, rd, t0, t1)
        assert (prim[0] == 0) == hit and prim[0] in (0, -1), (o, d, t0, t1, prim[0])
        if hit:
            assert gt[0] == t and tuple(gp[0]) == p and tuple(gn[0]) == tuple(float(x) for x in n) and bool(gff[0]) == ff, (o, d, gt, gp, gn, gff)


def _bary64(tri, ro, rd):
    """float64 ray / triangle-plane intersection: t and barycentric (u, v) of the plane point."""
    v0, v1, v2 = (np.array(tri[3 * k:3 * k + 3], np.float64) for k in range(3))
    e1, e2 = v1 - v0, v2 - v0
    n = np.cross(e1, e2)
    den = rd @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((v0 - ro) @ n) / den
        p = ro + t[:, None] * rd - v0
        d11, d12, d22 = e1 @ e1, e1 @ e2, e2 @ e2
        b1, b2 = p @ e1, p @ e2
        det = d11 * d22 - d12 * d12
        u = (d22 * b1 - d12 * b2) / det
        v = (d11 * b2 - d12 * b1) / det
    return t, u, v, n / np.linalg.norm(n)


def check_general_triangles(hit_rays, seed=5, n_tri=40, n_ray=400):
    """Random triangles, one per scene, rays aimed at points clearly inside / clearly outside: hit or miss as the
    float64 geometry says, t within FP32 rounding of the float64 t, hit point on the plane and inside the triangle,
    normal = +-unit normal facing the ray. Also: whatever is reported as a hit anywhere (grazing, near-in-plane and
    far-away rays included) lies in the triangle's plane and within its bounding sphere."""
    rng = np.random.default_rng(seed)
    for _ in range(n_tri):
        v = (rng.uniform(-4, 4, (3, 3)) * rng.choice([0.1, 1.0, 10.0])).astype(np.float32)
        tri = tuple(float(x) for x in v.reshape(-1))
        sc = _scene([scenes.Shape("triangle", tri, 0)])
        v64 = v.astype(np.float64)
        w = rng.dirichlet((1, 1, 1), n_ray)
        inside = rng.random(n_ray) < 0.5
        w[~inside] = w[~inside] * np.array([1.0, 1.0, -1.0]) + np.array([0.0, 0.0, -0.05])      # third weight negative: outside
        w /= w.sum(1, keepdims=True)
        target = w @ v64
        scale = np.abs(v64).max()
        ro = (target + rng.standard_normal((n_ray, 3)) * scale * rng.choice([0.5, 3.0, 50.0], (n_ray, 1))).astype(np.float32)
        rd = ((target - ro.astype(np.float64)) * rng.choice([0.25, 1.0, 7.0], (n_ray, 1))).astype(np.float32)
        # a fifth of the rays: nearly in the triangle's plane, from far away (the spurious-hit regime of FP32 Moeller-Trumbore)
        k = n_ray // 5
        e1, e2 = v64[1] - v64[0], v64[2] - v64[0]
        nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm)
        inplane = e1 / np.linalg.norm(e1)
        ro[:k] = (v64[0] - inplane * scale * rng.uniform(50, 5000, (k, 1)) + nrm * scale * rng.uniform(-1e-4, 1e-4, (k, 1))).astype(np.float32)
        rd[:k] = (inplane + nrm * rng.uniform(-1e-6, 1e-6, (k, 1))).astype(np.float32)
        prim, t, p, n, ff = hit_rays(sc, ro, rd, 0.001, INF)
        t64, u, vv, n64 = _bary64(tri, ro.astype(np.float64), rd.astype(np.float64))
        m = 1e-3
        sure_in = (u > m) & (vv > m) & (u + vv < 1 - m) & (t64 > 0.01) & (np.arange(n_ray) >= k)
        sure_out = ((u < -m) | (vv < -m) | (u + vv > 1 + m) | (t64 < 0)) & np.isfinite(t64) & (np.arange(n_ray) >= k)
        assert (prim[sure_in] == 0).all() and (prim[sure_out] == -1).all()
        assert sure_in.sum() > n_ray // 5 and sure_out.sum() > n_ray // 5
        # t against the float64 t: FP32 Moeller-Trumbore loses accuracy with the distance of the origin and with the
        # grazing angle, so the bound is on the error measured ACROSS the plane, relative to the size of the set-up
        reach = scale + np.linalg.norm(ro.astype(np.float64) - target, axis=1)
        err = np.abs(t - t64) * np.abs(rd.astype(np.float64) @ n64)
        assert (err[sure_in] <= 2e-5 * reach[sure_in]).all(), float((err[sure_in] / reach[sure_in]).max())
        hit = prim == 0
        # every reported hit: on the plane (to FP32 accuracy of the hit point), inside the minimal enclosing sphere + 2 %
        c = np.array([(v64[a] + v64[b]) / 2 for a, b in ((0, 1), (0, 2), (1, 2))])
        r_enc = max(np.linalg.norm(v64 - v64.mean(0), axis=1).max(), 0) * 1.3   # generous: centroid sphere x 1.3 contains the minimal sphere + margins
        ph = p[hit].astype(np.float64)
        assert (np.linalg.norm(ph - v64.mean(0), axis=1) <= r_enc).all()
        far = np.linalg.norm(ro[hit].astype(np.float64) - ph, axis=1)
        assert (np.abs((ph - v64[0]) @ n64) <= 1e-5 * (scale + far)).all()
        front = (rd[hit].astype(np.float64) @ n64) < 0
        sel = np.abs(rd[hit].astype(np.float64) @ n64) > 1e-4 * np.linalg.norm(rd[hit], axis=1)
        assert (ff[hit].astype(bool) == front)[sel].all()
        assert np.allclose(n[hit], np.where(ff[hit].astype(bool)[:, None], n64, -n64), atol=2e-6)


def check_triangle_pair_equals_rectangle(hit_rays, seed=9, n=3000):
    """Two triangles that tile a `rectangle` (shapes.h:161-192) against the rectangle itself on the same rays."""
    x0, x1, y0, y1, k = -1.0, 3.0, 0.5, 2.5, -3.0     # power-of-two edge lengths: the triangles' determinants have exact reciprocals
    rect = _scene([scenes.Shape("xy_rect", (x0, x1, y0, y1, k), 0)])
    tris = _scene([scenes.Shape("triangle", (x0, y0, k, x1, y0, k, x0, y1, k), 0),
                   scenes.Shape("triangle", (x1, y0, k, x1, y1, k, x0, y1, k), 0)])
    rng = np.random.default_rng(seed)
    # (a) axis-parallel rays through dyadic points, incl. the rectangle's border and the shared diagonal: exact arithmetic
    gx = x0 + (x1 - x0) * rng.integers(-2, 19, n) / 16.0
    gy = y0 + (y1 - y0) * rng.integers(-2, 19, n) / 16.0
    ro = np.stack([gx, gy, rng.choice([1.0, -7.0, 2.5], n)], 1).astype(np.float32)
    rd = np.stack([np.zeros(n), np.zeros(n), rng.choice([-1.0, 1.0, -0.5, 4.0], n)], 1).astype(np.float32)
    pa, ta, ppa, na, fa = hit_rays(rect, ro, rd, 0.001, INF)
    pb, tb, ppb, nb, fb = hit_rays(tris, ro, rd, 0.001, INF)
    assert np.array_equal(pa >= 0, pb >= 0) and 0.2 < (pa >= 0).mean() < 0.9
    assert np.array_equal(ta, tb) and np.array_equal(ppa, ppb) and np.array_equal(na, nb) and np.array_equal(fa, fb)
    # (b) oblique rays aimed clearly inside (away from the border and the diagonal) or clearly outside
    tx, ty = rng.uniform(x0 - 1, x1 + 1, n), rng.uniform(y0 - 1, y1 + 1, n)
    ro = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    rd = (np.stack([tx, ty, np.full(n, k)], 1) - ro).astype(np.float32) * rng.choice([0.3, 1.0, 5.0], (n, 1)).astype(np.float32)
    uu, vv = (tx - x0) / (x1 - x0), (ty - y0) / (y1 - y0)
    m = 1e-3
    clear = ((uu > m) & (uu < 1 - m) & (vv > m) & (vv < 1 - m) & (np.abs(uu + vv - 1) > m)) | (uu < -m) | (uu > 1 + m) | (vv < -m) | (vv > 1 + m)
    clear &= np.abs(ro[:, 2] - k) > 0.05
    pa, ta, ppa, na, fa = hit_rays(rect, ro, rd, 0.001, INF)
    pb, tb, ppb, nb, fb = hit_rays(tris, ro, rd, 0.001, INF)
    h = (pa >= 0) & clear
    assert np.array_equal((pa >= 0)[clear], (pb >= 0)[clear]) and h.sum() > n // 10
    assert np.allclose(ta[h], tb[h], rtol=1e-5, atol=0) and np.allclose(ppa[h], ppb[h], rtol=1e-5, atol=1e-5)
    assert np.array_equal(na[h], nb[h]) and np.array_equal(fa[h], fb[h])


def check_t_min_that_is_no_float(hit_rays):
    """40 stacked unit triangles at z = 0, -1, -2, ... and rays straight down from z = 1: hits at t = 1, 2, 3, ... exactly.
    With t_min one part in 2^30 above 1 (a double that is no float: it rounds DOWN to 1.0f) the hit at t = 1 is outside the range
    and the triangle behind it, at t = 2, must be reported (the triangle queue's workers test against the owner's range
    as floats; a hit at rd(t_min) must not hide the valid one behind it). 40 triangles: enough for the filter group and its queue."""
    shapes = [scenes.Shape("triangle", (0.0, 0.0, -float(k), 1.0, 0.0, -float(k), 0.0, 1.0, -float(k)), 0) for k in range(40)]
    sc = _scene(shapes)
    n = 70                                                             # more rays than a wave
    ro = np.tile(np.array([[0.25, 0.25, 1.0]], np.float32), (n, 1))
    ro[:, 0] += np.arange(n, dtype=np.float32) / 256.0                 # exact in FP32, all inside the triangles
    rd = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (n, 1))
    for t_min, t_max, want_prim, want_t in ((1.0 + 2.0 ** -30, INF, 1, 2.0), (1.0, INF, 0, 1.0), (3.0 - 2.0 ** -30, INF, 2, 3.0),
                                            (2.0 + 2.0 ** -40, 2.5, -1, 0.0), (1.0 + 2.0 ** -30, 2.0, 1, 2.0)):
        prim, t, p, nrm, ff = hit_rays(sc, ro, rd, t_min, t_max)
        assert (prim == want_prim).all(), (t_min, t_max, prim[:4])
        if want_prim >= 0:
            assert (t == want_t).all(), (t_min, t[:4])
