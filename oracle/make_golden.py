#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY. Generates the golden fixtures under tests/golden/ by running the
*reference's own code* (oracle/_ref/ref_harness = /root/reference/{calc,hitPosition}.cpp compiled
unmodified by oracle/build_ref.sh) in this container. The reference cannot travel to the GPU box,
so the vectors are committed as compressed .npz files (data only: inputs and expected outputs).

    ./oracle/build_ref.sh && python oracle/make_golden.py [--only NAME]

Fixtures:
  rng.npz, vec3.npz, camera.npz, shapehits.npz, materials.npz, medium.npz  -- unit-level known answers
  primary_<scene>_<W>x<H>.npz   -- pixel-centre primary-hit ids (+ box side, t hash, t subsample)
  render_<scene>_<W>x<H>_s<spp>_d<depth>.npz -- seeded sequence-exact radiance sums (RefRng 1,2,3)
  mean_builtin_160x90.npz       -- 4096-spp converged means for statistical comparison
Each scene used is stored beside the fixture as scene_<name>.txt (the harness input).
"""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from ray_amd import scenes  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
_DT = {"f4": np.float32, "f8": np.float64, "i4": np.int32, "u1": np.uint8}


def read_rtfx(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            parts = line.decode().split()
            name, dt, nd = parts[0], _DT[parts[1]], int(parts[2])
            shape = tuple(int(x) for x in parts[3:3 + nd])
            n = int(np.prod(shape)) if shape else 1
            out[name] = np.frombuffer(f.read(n * np.dtype(dt).itemsize), dtype=dt).reshape(shape).copy()
    return out


def run(*args):
    cmd = [HARNESS] + [str(a) for a in args]
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def save(name, arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def scene_file(sc, tmp):
    txt = sc.to_text()
    with open(os.path.join(GOLD, f"scene_{sc.name}.txt"), "w") as f:
        f.write(txt)
    p = os.path.join(tmp, f"{sc.name}.txt")
    with open(p, "w") as f:
        f.write(txt)
    return p


def golden_scenes():
    """name -> Scene for the goldens: aperture 0 for primary maps (no RNG in the ray)."""
    b0 = scenes.builtin_scene(aperture=0.0); b0.name = "builtin_a0"
    r0 = scenes.random_scene(11, 1, aperture=0.0); r0.name = "random11_a0"
    x0 = scenes.box_scene(aperture=0.0); x0.name = "box_a0"
    b1 = scenes.builtin_scene(aperture=0.1); b1.name = "builtin"
    r1 = scenes.random_scene(3, 1, aperture=0.1); r1.name = "random3"
    x1 = scenes.box_scene(aperture=0.05); x1.name = "box_a005"
    xn = scenes.box_scene(aperture=0.05, light_on_panel=True); xn.name = "box_nanlight"
    bxy = scenes.builtin_scene(light="xy"); byz = scenes.builtin_scene(light="yz")
    return dict(builtin_a0=b0, random11_a0=r0, box_a0=x0, builtin=b1, random3=r1, box_a005=x1, box_nanlight=xn,
                builtin_xylight=bxy, builtin_yzlight=byz)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--missing", action="store_true", help="only write fixtures that do not exist yet")
    args = ap.parse_args()
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness missing: run oracle/build_ref.sh first")
    os.makedirs(GOLD, exist_ok=True)
    want = lambda n: args.only is None or args.only == n  # noqa: E731
    have = lambda name: args.missing and os.path.exists(os.path.join(GOLD, name + ".npz"))  # noqa: E731
    S = golden_scenes()
    with tempfile.TemporaryDirectory() as tmp:
        for unit in ("rng", "vec3", "camera", "shapehits", "materials", "medium"):
            if want(unit) and not have(unit):
                out = os.path.join(tmp, unit + ".rtfx")
                run(unit, out)
                save(unit, read_rtfx(out))
        if want("primary"):
            for sname, W, H in (("builtin_a0", 640, 360), ("builtin_a0", 1920, 1080),
                                ("random11_a0", 640, 360), ("box_a0", 640, 360),
                                ("random11_a0", 3840, 2160)):        # BASELINE.json configs[2] at its full frame size
                if have(f"primary_{sname}_{W}x{H}"):
                    continue
                sp = scene_file(S[sname], tmp)
                out = os.path.join(tmp, "p.rtfx")
                run("primary", sp, W, H, out)
                d = read_rtfx(out)
                t = d.pop("t")
                d["t_sha256"] = np.frombuffer(hashlib.sha256(t.tobytes()).digest(), dtype=np.uint8)
                d["t_sub"] = t[::9, ::9].copy()           # every 9th pixel, full double bits
                save(f"primary_{sname}_{W}x{H}", d)
        if want("render"):
            for sname, W, H, spp, depth in (("builtin", 64, 36, 4, 8), ("random3", 48, 27, 2, 10),
                                            ("box_a005", 48, 27, 2, 6), ("box_nanlight", 32, 18, 2, 6),
                                            ("builtin_xylight", 48, 27, 4, 8), ("builtin_yzlight", 48, 27, 4, 8)):
                if have(f"render_{sname}_{W}x{H}_s{spp}_d{depth}"):
                    continue
                sp = scene_file(S[sname], tmp)
                out = os.path.join(tmp, "r.rtfx")
                run("render", sp, W, H, spp, depth, 1, 2, 3, out)
                save(f"render_{sname}_{W}x{H}_s{spp}_d{depth}", read_rtfx(out))
        if want("mean") and not have("mean_builtin_160x90"):
            sp = scene_file(S["builtin"], tmp)
            out = os.path.join(tmp, "m.rtfx")
            run("render", sp, 160, 90, 4096, 8, 11, 12, 13, out)
            save("mean_builtin_160x90", read_rtfx(out))


if __name__ == "__main__":
    main()
