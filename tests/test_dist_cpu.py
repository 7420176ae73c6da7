"""Multi-process path on CPU: the frame split of ray_amd/dist.py with world_size 2 and 3 over gloo."""
import os
import subprocess
import sys

import pytest

from ray_amd import dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_splits_tile_the_range():
    for n in (0, 1, 7, 64, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for r in range(world):
                b, c = rdist.spp_split(n, r, world)
                assert b == pos and c >= 0
                pos += c
            assert pos == n
            assert [rdist.row_split(n, r, world) for r in range(world)] == [rdist.spp_split(n, r, world) for r in range(world)]
    assert rdist.spp_split(256, 3, 8) == (96, 32)
    with pytest.raises(ValueError):
        rdist.spp_split(8, 2, 2)
    p = rdist.plan("rows", 64, 36, 16, 1, 4)
    assert p == dict(y_begin=9, y_count=9, band_first=0, band_stride=1, spp_begin=0, spp_count=16, zero_first=True)
    p = rdist.plan("tile", 64, 36, 16, 1, 4)
    assert p == dict(y_begin=0, y_count=36, band_first=1, band_stride=4, spp_begin=0, spp_count=16, zero_first=False)
    assert [rdist.padded_rows(h, w) for h, w in ((36, 4), (2160, 8), (27, 2), (8, 1), (1, 3))] == [64, 2176, 32, 8, 24]
    assert rdist.plan("tile", 64, 36, 16, 0, 1)["zero_first"] is False
    with pytest.raises(ValueError):
        rdist.plan("columns", 1, 1, 1, 0, 1)
    for H in (1, 8, 27, 36, 2160):                      # the ranks' bands tile [0, H) exactly once
        for world in (1, 2, 3, 8):
            rows = sorted(r for k in range(world) for r in rdist.band_rows(H, k, world))
            assert rows[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(rows, rows[1:])) and rows[-1][0] + rows[-1][1] == H


@pytest.mark.parametrize("world", [2, 3])
def test_split_frame_over_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29611 + world
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
