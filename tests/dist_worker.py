"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job. Renders its share of a small
frame with the CPU oracle (test infrastructure: it stands in for the device here), sums the
accumulators with ray_amd.dist.reduce_accumulators and lets rank 0 compare with the whole frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from ray_amd import dist as rdist, scenes  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp, depth, seed = 48, 27, 6, 6, 0xC0FFEE
    sc = scenes.get_scene("random", half=2)
    ok = True
    for split in ("spp", "tile", "rows"):
        p = rdist.plan(split, W, H, spp, rank, world)
        acc = torch.zeros((H, W, 4), dtype=torch.float32)
        rays = 0
        bands = rdist.band_rows(H, p["band_first"], p["band_stride"]) if p["band_stride"] > 1 else [(p["y_begin"], p["y_count"])]
        for y0, yc in bands:                              # the oracle renders one contiguous band per call
            part, n = O.render(sc, W, H, p["spp_count"], depth, rng_mode=1, seed64=seed, math_mode=1,
                               y_begin=y0, y_count=yc, spp_begin=p["spp_begin"])
            acc += torch.from_numpy(part)
            rays += n
        rdist.reduce_accumulators(acc, world)
        r = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(r)
        if rank == 0:
            full, rays_full = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=seed, math_mode=1)
            got = acc.numpy()
            assert int(r.item()) == rays_full, (int(r.item()), rays_full)
            assert (got[..., 3] == spp).all()
            if split != "spp":
                ok &= np.array_equal(got, full, equal_nan=True)          # disjoint bands + zeros: bit-exact
            else:
                ok &= np.allclose(got[..., :3], full[..., :3], rtol=2e-6, atol=1e-6)   # (a+b)+(c+d) vs sequential
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
