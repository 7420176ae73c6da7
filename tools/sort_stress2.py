#!/usr/bin/env python3
"""Repeats tests/test_gpu_parity.py::test_block_sorted_shading_gives_the_same_frame in one process (optionally after the whole suite)."""
import os, sys, traceback
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
if len(sys.argv) > 1:
    print("pytest rc", pytest.main(["tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", sys.argv[1], "--timeout", "120"]), flush=True)
import test_gpu_parity as T  # noqa: E402
from ray_amd import api  # noqa: E402
ctx = api.Context(0)
bad = 0
for rep in range(12):
    try:
        T.test_block_sorted_shading_gives_the_same_frame(ctx)
    except AssertionError:
        bad += 1
        print("rep", rep, "FAILED:", traceback.format_exc().splitlines()[-1], flush=True)
print("bad", bad, "of 12", flush=True)
ctx.close()
