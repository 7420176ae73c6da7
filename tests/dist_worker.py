"""Worker of the multi-process tests: one rank of a world_size-N gloo job renders its share of a small frame,
exchanges it with ray_amd.dist (all-reduce for spp / rows splits, band all-gather for the tile split) and rank 0
compares with the whole frame.

    dist_worker.py            CPU (tests/test_dist_cpu.py): the oracle stands in for the device -- test infrastructure
                              only, it checks the host-side split / exchange logic where there is no GPU.
    dist_worker.py --device   GPU box (tests/test_gpu_parity.py): every rank drives the HIP library on cuda:0 through
                              the C ABI, torch tensors as accumulators on torch's stream (rt_set_stream), gloo moves them;
                              the whole frame it is compared with comes from one rank's plain rt_render AND the oracle."""
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
    on_device = "--device" in sys.argv
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp, depth, seed = 48, 27, 6, 6, 0xC0FFEE
    sc = scenes.get_scene("random", half=2)
    dev = torch.device("cpu")
    ctx = None
    if on_device:
        from ray_amd import api
        W, H, spp = 200, 77, 8
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        ctx = api.Context(0)
        ctx.upload_scene(sc)
        ctx.set_stream(torch.cuda.current_stream())
    ok = True
    for split in ("spp", "tile", "rows"):
        p = rdist.plan(split, W, H, spp, rank, world)
        rows = rdist.padded_rows(H, world) if split == "tile" else H
        # garbage first: a share that relies on zeros it did not write, or a kernel that runs before the fill, shows
        acc = torch.full((rows, W, 4), float("nan"), dtype=torch.float32, device=dev)
        if p["zero_first"] or split == "spp":
            acc.zero_()
        rays = 0
        if on_device:
            r_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            ctx.render_into(W, H, acc, p["spp_count"], depth, seed=seed, y_begin=p["y_begin"], y_count=p["y_count"],
                            spp_begin=p["spp_begin"], rays=r_dev, band_first=p["band_first"], band_stride=p["band_stride"])
            rays = int(r_dev.item())
        else:
            bands = rdist.band_rows(H, p["band_first"], p["band_stride"]) if p["band_stride"] > 1 else [(p["y_begin"], p["y_count"])]
            part = np.zeros((H, W, 4), np.float32)
            for y0, yc in bands:                              # the oracle renders one contiguous band per call
                _, n = O.render(sc, W, H, p["spp_count"], depth, rng_mode=1, seed64=seed, math_mode=1,
                                y_begin=y0, y_count=yc, spp_begin=p["spp_begin"], accum=part)
                rays += n
                acc[y0:y0 + yc] = torch.from_numpy(part[y0:y0 + yc])
        mine = acc.clone() if split == "spp" else None
        if split == "tile":
            rdist.gather_bands(acc, rank, world)
        else:
            rdist.reduce_accumulators(acc, world)
        r = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(r)
        got = acc[:H].cpu().numpy()
        full, rays_full = O.render(sc, W, H, spp, depth, rng_mode=1, seed64=seed, math_mode=1, n_threads=4)
        good = int(r.item()) == rays_full and bool((got[..., 3] == spp).all())
        if split != "spp":
            good &= np.array_equal(got, full, equal_nan=True)          # disjoint bands: bit-exact on every rank
        else:
            good &= np.allclose(got[..., :3], full[..., :3], rtol=2e-6, atol=1e-6, equal_nan=True)   # (a+b)+(c+d) vs sequential
        if split == "spp":
            # the sum in rank order: the same bits on every rank, and those of ((share 0 + share 1) + share 2) ... from the oracle
            rdist.reduce_accumulators(mine, world, ordered=True)
            want = None
            for k in range(world):
                q = rdist.plan("spp", W, H, spp, k, world)
                share, _ = O.render(sc, W, H, q["spp_count"], depth, rng_mode=1, seed64=seed, math_mode=1, spp_begin=q["spp_begin"])
                want = share if want is None else want + share
            good &= np.array_equal(mine.cpu().numpy().view(np.uint32), want.view(np.uint32))
        if on_device and rank == 0:
            one, rays_one = ctx.render(W, H, spp, depth, seed=seed)     # the same frame from one plain rt_render
            good &= rays_one == rays_full and np.array_equal(one, full, equal_nan=True)
        if not good:
            print(f"rank {rank}: split {split} differs (rays {int(r.item())} vs {rays_full})", flush=True)
        ok &= good
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    if ctx is not None:
        ctx.close()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
