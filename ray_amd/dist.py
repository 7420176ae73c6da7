"""Multi-GPU split of one frame, one process per GPU under torch.distributed.

Every pixel-sample is independent (the counter RNG is keyed by (seed, pixel, sample)), so a frame is
sharded with no data-path exchange except ONE sum of the FP32 accumulators at the end:
  * spp-split: rank r renders samples [spp*r/N, spp*(r+1)/N) of every pixel;
  * tile-split: rank r renders all samples of the 8-row bands r, r + N, r + 2N, ... into a zeroed
    accumulator (interleaved, so that every rank gets the same mix of sky and geometry rows);
  * rows-split: the same with one contiguous band [H*r/N, H*(r+1)/N) per rank.
Both end with all_reduce(sum) over the [H][W][4] accumulator (RCCL over xGMI with backend "nccl",
gloo on CPU in the tests). The functions here are pure host logic: they are what bench.py runs and
what tests/test_dist_cpu.py exercises with world_size 2 on CPU."""
from __future__ import annotations


def spp_split(spp: int, rank: int, world: int):
    """Sample range of `rank`: (begin, count). Ranges tile [0, spp) in rank order."""
    if world < 1 or not 0 <= rank < world or spp < 0:
        raise ValueError(f"bad split spp={spp} rank={rank} world={world}")
    b, e = spp * rank // world, spp * (rank + 1) // world
    return b, e - b


def row_split(height: int, rank: int, world: int):
    """Row band of `rank`: (y_begin, y_count). Bands tile [0, height) in rank order."""
    if world < 1 or not 0 <= rank < world or height < 0:
        raise ValueError(f"bad split height={height} rank={rank} world={world}")
    b, e = height * rank // world, height * (rank + 1) // world
    return b, e - b


def plan(split: str, W: int, H: int, spp: int, rank: int, world: int) -> dict:
    """What `rank` renders: rows [y_begin, +y_count) or, when band_stride > 1, the 8-row bands band_first,
    band_first + band_stride, ... of the whole frame; samples [spp_begin, +spp_count); and whether its
    accumulator must be zeroed first (image splits: the other ranks' rows must contribute zeros)."""
    if split not in ("spp", "tile", "rows"):
        raise ValueError(f"unknown split {split!r}")
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad split rank={rank} world={world}")
    if split == "spp" or world == 1:
        s0, sc = spp_split(spp, rank, world)
        return dict(y_begin=0, y_count=H, band_first=0, band_stride=1, spp_begin=s0, spp_count=sc, zero_first=False)
    if split == "tile":
        return dict(y_begin=0, y_count=H, band_first=rank, band_stride=world, spp_begin=0, spp_count=spp, zero_first=True)
    y0, yc = row_split(H, rank, world)
    return dict(y_begin=y0, y_count=yc, band_first=0, band_stride=1, spp_begin=0, spp_count=spp, zero_first=True)


BAND_ROWS = 8   # rows per band of rt_render_bands (one tile row of the render kernel)


def band_rows(H: int, band_first: int, band_stride: int):
    """The (y_begin, y_count) row ranges of the bands band_first, band_first + band_stride, ... of an H-row frame."""
    out = []
    for b in range(band_first, (H + BAND_ROWS - 1) // BAND_ROWS, band_stride):
        out.append((b * BAND_ROWS, min(BAND_ROWS, H - b * BAND_ROWS)))
    return out


def reduce_accumulators(accum, world: int):
    """Sum the per-rank accumulators in place on every rank (no-op for one rank)."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(accum)
    return accum
