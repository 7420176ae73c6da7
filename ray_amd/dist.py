"""Multi-GPU split of one frame, one process per GPU under torch.distributed.

Every pixel-sample is independent (the counter RNG is keyed by (seed, pixel, sample)), so a frame is
sharded with no data-path exchange except ONE sum of the FP32 accumulators at the end:
  * spp-split: rank r renders samples [spp*r/N, spp*(r+1)/N) of every pixel;
  * tile-split: rank r renders all samples of the 8-row bands r, r + N, r + 2N, ... (interleaved, so that
    every rank gets the same mix of sky and geometry rows); the disjoint bands are exchanged by ONE
    all-gather (gather_bands), no zero fill and no sum;
  * rows-split: one contiguous band [H*r/N, H*(r+1)/N) per rank into a zeroed accumulator, then all-reduce.
spp-split ends with all_reduce(sum) over the [H][W][4] accumulator (RCCL over xGMI with backend "nccl",
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
        return dict(y_begin=0, y_count=H, band_first=rank, band_stride=world, spp_begin=0, spp_count=spp, zero_first=False)
    y0, yc = row_split(H, rank, world)
    return dict(y_begin=y0, y_count=yc, band_first=0, band_stride=1, spp_begin=0, spp_count=spp, zero_first=True)


BAND_ROWS = 8   # rows per band of rt_render_bands (one tile row of the render kernel)


def band_rows(H: int, band_first: int, band_stride: int):
    """The (y_begin, y_count) row ranges of the bands band_first, band_first + band_stride, ... of an H-row frame."""
    out = []
    for b in range(band_first, (H + BAND_ROWS - 1) // BAND_ROWS, band_stride):
        out.append((b * BAND_ROWS, min(BAND_ROWS, H - b * BAND_ROWS)))
    return out


def reduce_accumulators(accum, world: int, ordered: bool = False):
    """spp-split: sum the per-rank accumulators in place on every rank (no-op for one rank).
    ordered=False: one all-reduce; the order of the additions is the collective's (ring or tree, it may change with the
    library version and the world size), so the low bits of a pixel can differ from run to run of different set-ups.
    ordered=True ("reduce in a fixed rank order for bit-reproducibility"): one all-gather of the whole
    accumulators (world times the bytes) and ((a0 + a1) + a2) + ... in rank order on every rank - the same bits on every
    rank, whatever the collective does, and the bits ray_amd/host/render_multi.cpp produces in one process."""
    if world > 1:
        import torch
        import torch.distributed as dist
        if not ordered:
            dist.all_reduce(accum)
            return accum
        mine = accum.contiguous()
        parts = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(parts.view(-1), mine.view(-1))
        accum.copy_(parts[0])
        for r in range(1, world):
            accum.add_(parts[r])
    return accum


def padded_rows(H: int, world: int) -> int:
    """Rows of an accumulator that gather_bands can exchange: H rounded up to a whole number of bands per rank."""
    nb = (H + BAND_ROWS - 1) // BAND_ROWS
    return ((nb + world - 1) // world) * world * BAND_ROWS


def gather_bands(accum, rank: int, world: int):
    """tile-split: every rank has rendered the bands rank, rank + world, ... of `accum` ([padded_rows(H, world)][W][4],
    rows >= H unused); afterwards every rank holds all bands. The bands are disjoint, so this is an all-gather of
    each rank's 1/world of the frame -- no zero fill, no sum, half the bytes of an all-reduce."""
    if world == 1:
        return accum
    import torch
    import torch.distributed as dist
    rows, W, C = accum.shape
    per = rows // (world * BAND_ROWS)
    if per * world * BAND_ROWS != rows:
        raise ValueError(f"accumulator has {rows} rows, not a whole number of {BAND_ROWS}-row bands per rank (use padded_rows)")
    a = accum.view(per, world, BAND_ROWS, W, C)            # band b = k * world + r  ->  a[k, r]
    mine = a[:, rank].contiguous()
    out = torch.empty((world, per, BAND_ROWS, W, C), dtype=accum.dtype, device=accum.device)
    dist.all_gather_into_tensor(out.view(-1), mine.view(-1))     # flat: gloo and RCCL both concatenate along dim 0
    a.copy_(out.permute(1, 0, 2, 3, 4))
    return accum
