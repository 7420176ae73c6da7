#!/usr/bin/env python3
"""Runs the GPU parity tests in this process, then repeats renders with block-sorted shading and compares ray counts and frames."""
import os, sys, zlib
import numpy as np
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
rc = pytest.main(["tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", sys.argv[1] if len(sys.argv) > 1 else "not block_sorted", "--timeout", "120"])
print("pytest rc", rc, flush=True)
from ray_amd import api, scenes  # noqa: E402
ctx = api.Context(0)
w, h, n, depth = 1280, 720, 16, 8
ctx.upload_scene(scenes.get_scene("builtin"))
want = None
bad = 0
for rep in range(60):
    mode = rep & 1
    ctx.set_shading_sort(mode)
    buf = ctx.alloc(w * h * 16).zero()
    r = ctx.alloc(8).zero()
    ctx.render_into(w, h, buf, n, depth, seed=9, rays=r, band_first=1, band_stride=3)
    got = (int(r.to_numpy(np.uint64, (1,))[0]), zlib.crc32(buf.to_numpy(np.uint32, (h, w, 4)).tobytes()))
    if want is None:
        want = got
    if got != want:
        bad += 1
        print("DIFF rep", rep, "mode", mode, got, want, flush=True)
    buf.free(); r.free()
print("bad", bad, "of 60", flush=True)
ctx.close()
