#!/usr/bin/env python3
"""Benchmark of the ray path on MI355X: one "step" = one full frame of the named workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5]

N > 1 is launched by torch.distributed.run (one rank per GPU): the samples-per-pixel of every
pixel are split over the ranks (spp-split) and the FP32 accumulators are summed by
an RCCL all-reduce over xGMI inside the timed region (c5: image bands are split and exchanged by one
all-gather) -> "scaling": "strong".

Workloads (BASELINE.json configs):
  c3  random-spheres scene (~485 spheres), 3840x2160, 256 spp, depth 8   <- default (the frame
      BASELINE.json's north_star target names; it fits one GPU)
  c2  built-in 5-primitive scene, 1920x1080, 64 spp, depth 10
  c4  ~10k-triangle mesh, 1920x1080, 64 spp, depth 8
  c5  mixed spheres + triangles, 3840x2160, 1024 spp (tile-split when N > 1)

Rank 0 prints ONE JSON line: metric Mrays/s (closest-hit scans per second, primary + bounce rays),
ms_per_step = ms per frame, plus `roofline` (FP32 VALU, algorithmic flops of the closest-hit tests
over the kernel time measured with HIP events on the library's stream) and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (vector)"
FLOPS_SPHERE, FLOPS_RECT, FLOPS_TRI = 17, 8, 45   # SURVEY.md section 8d, per ray-primitive test

WORKLOADS = {
    "c3": dict(scene="random", kw=dict(half=11), W=3840, H=2160, spp=256, depth=8, split="spp",
               desc="random-spheres 3840x2160 256spp depth8"),
    "c2": dict(scene="builtin", kw={}, W=1920, H=1080, spp=64, depth=10, split="spp",
               desc="built-in sphere scene 1920x1080 64spp depth10"),
    "c4": dict(scene="tris", kw=dict(n=71), W=1920, H=1080, spp=64, depth=8, split="spp",
               desc="10k-triangle mesh 1920x1080 64spp depth8"),
    "box": dict(scene="box", kw={}, W=1920, H=1080, spp=64, depth=10, split="spp",
                desc="box_scene (rayview.cpp:262-378, 169 flattened primitives) 1920x1080 64spp depth10"),
    "c5": dict(scene="mixed", kw=dict(half=11, n=71), W=3840, H=2160, spp=1024, depth=8, split="tile",
               desc="mixed spheres+triangles 3840x2160 1024spp depth8"),
}


def flops_per_ray(flat):
    k = flat["prim_kind"]
    return int((k == 0).sum()) * FLOPS_SPHERE + int(((k >= 1) & (k <= 3)).sum()) * FLOPS_RECT + int((k == 4).sum()) * FLOPS_TRI


def cpu_baseline(scene, wl, budget_s=12.0):
    """Time the CPU side on a bounded sample of the same workload (same scene, camera, depth; a
    reduced frame). Preferred: the reference itself (oracle/_ref/ref_harness, built in the build
    container from the unmodified reference sources; single-threaded by construction). Otherwise the
    oracle port (oracle/librt_oracle.so), also on one thread."""
    # sized for roughly 10-20 s of single-core work per leg on the reference / port
    sw, sh, sspp = 640, 360, 12
    if wl["scene"] == "builtin":
        sspp = 64          # 4x BASELINE.json configs[0] (640x360, 16 spp)
    elif wl["scene"] in ("tris", "mixed"):
        sw, sh, sspp = 160, 90, 2
    out = {}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    # port, 1 thread
    t0 = time.perf_counter()
    _, rays = O.render(scene, sw, sh, sspp, wl["depth"], rng_mode=1, seed64=1, math_mode=1, n_threads=1)
    dt = time.perf_counter() - t0
    port = dict(value=rays / dt / 1e6, unit="Mrays/s", cores=1, kind="port",
                sample=f"{wl['scene']} scene {sw}x{sh} {sspp}spp depth{wl['depth']} ({rays} rays, {dt:.2f}s)")
    # port, all the cores this job may use
    ncore = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    _, rays_mt = O.render(scene, sw, sh, sspp * 8, wl["depth"], rng_mode=1, seed64=1, math_mode=1, n_threads=ncore)
    dt_mt = time.perf_counter() - t0
    port["all_cores"] = dict(value=rays_mt / dt_mt / 1e6, unit="Mrays/s", cores=ncore)
    out = port
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    has_tri = any(s.kind == "triangle" for s in scene.world)
    if os.path.exists(harness) and not has_tri:
        try:
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                sp = os.path.join(tmp, "scene.txt")
                open(sp, "w").write(scene.to_text())
                t0 = time.perf_counter()
                r = subprocess.run([harness, "render", sp, str(sw), str(sh), str(sspp), str(wl["depth"]), "1", "2", "3"],
                                   capture_output=True, text=True, timeout=300)
                dt_ref = time.perf_counter() - t0
            if r.returncode == 0:
                # the harness prints "rays N" (its own count of hitFromList calls, taken by the linker's --wrap around the
                # unmodified reference) and "... X s (Y Msamples/s)"; an older binary without the count falls back to the
                # port's rays per sample (same workload, same estimator: they agree to <1%) and says so
                sec = float(r.stderr.split(":")[-1].split("s")[0])
                counted = [ln for ln in r.stderr.splitlines() if ln.startswith("rays ")]
                rays_ref = int(counted[0].split()[1]) if counted else rays
                out = dict(value=rays_ref / sec / 1e6, unit="Mrays/s", cores=1, kind="reference",
                           sample=f"{wl['scene']} scene {sw}x{sh} {sspp}spp depth{wl['depth']} ({rays_ref} rays, {sec:.2f}s)"
                           "; reference calc.cpp/hitPosition.cpp compiled unmodified (oracle/_ref), rays = "
                           + ("its own hitFromList calls" if counted else "rays/sample of the port run"),
                           port=dict(value=port["value"], cores=1, rays=rays), all_cores=port["all_cores"])
        except Exception as e:   # the reference binary is optional on the GPU box
            out["reference_error"] = str(e)[:200]
    return out


def executed_work(pmc, k_ms):
    """Executed (not algorithmic) VALU work of one render_kernel launch from a committed rocprofv3 PMC summary
    (profiles/r3_pmc.json, written by tools/prof.sh for exactly this configuration). How gfx950 counts, settled with
    tools/valu_bench under --pmc (profiles/r2_valu_bench_pmc.txt): SQ_INSTS_VALU_{FMA,MUL,ADD}_F32 count one per
    wave-instruction whether packed or not, so executed FP32 flops are only bounded: [every instruction unpacked,
    every instruction packed]; SQ_ACTIVE_INST_VALU counts issue slots of 4 cycles (FP64 fma/mul/add and conversions
    take one slot like FP32, v_rcp/rsq_f64 four, v_rcp/sqrt_f32 two); SQ_THREAD_CYCLES_VALU adds 64 per slot with all
    lanes on; SQ_INSTS_VALU_FLOPS_FP64 counts 2 per FP64 fma and 1 per mul/add/rcp/rsq."""
    c = pmc["counters_per_launch"]
    t = pmc["avg_ns"] * 1e-9
    fma, mul, add = c["SQ_INSTS_VALU_FMA_F32"], c["SQ_INSTS_VALU_MUL_F32"], c["SQ_INSTS_VALU_ADD_F32"]
    lo = (2 * fma + mul + add) * 64 / t / 1e12
    lane_util = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    return {
        "executed_tflops_range": [round(lo, 2), round(2 * lo, 2)],
        "executed_frac_range": [round(lo / FP32_VALU_PEAK_TFLOPS, 4), round(2 * lo / FP32_VALU_PEAK_TFLOPS, 4)],
        "lane_util": round(lane_util, 4),
        # issue slots used / available: 4 cycles per slot, 256 CUs x 4 SIMDs at 2.4 GHz for the profiled launch's duration
        "valu_issue_utilisation": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (t * 2.4e9 * 1024), 4),
        "fp64_issue_share_max": round(c["SQ_INSTS_VALU_FLOPS_FP64"] / c["SQ_ACTIVE_INST_VALU"], 4),
        "valu_insts_per_ray": None,
        "profile_kernel_ms": round(pmc["avg_ns"] * 1e-6, 3),
        "profile_vs_live_kernel_ms": round(pmc["avg_ns"] * 1e-6 / k_ms, 4) if k_ms > 0 else None,
        "source": "profiles/r3_pmc.json (rocprofv3 --pmc passes of this configuration; packed and unpacked FP32 instructions are "
                  "counted alike, hence a range)",
    }


def quick_run(ctx, api, scenes, torch, dev, name, frames=1):
    """One warm-up frame and `frames` timed frames of another workload on this GPU (the `also` dict)."""
    wl = WORKLOADS[name]
    W, H, spp, depth = wl["W"], wl["H"], wl["spp"], wl["depth"]
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    # warm-up: a frame of the same shape, so that the colour buffer of the frame's end (freed by rt_upload_scene, sized by the sample
    # count) is allocated before the timed frame; the 2.4 s frame of c5 warms up with a short one and carries its one hipMalloc
    ctx.render_into(W, H, acc, spp if name != "c5" else 8, depth, seed=0xC0FFEE, rays=rays)
    rays.zero_()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    k_ms = 0.0
    for _ in range(frames):
        ctx.render_into(W, H, acc, spp, depth, seed=0xC0FFEE, rays=rays)
        k_ms += ctx.last_kernel_ms()[0]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    n = int(rays.item())
    out = {"workload": wl["desc"], "ms_per_frame": round(1e3 * dt / frames, 3), "kernel_ms": round(k_ms / frames, 3),
           "Mrays_per_s": round(n / dt / 1e6, 1), "rays_per_frame": n // frames}
    if name == "c5":
        # the same frame the way a progressive caller renders it: 16 calls of 64 samples, each continuing the sums of the one before
        # (RT_RENDER_CONTINUE: the same bits as the one call above, compared here)
        ref = acc.clone()
        per = spp // 16
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(16):
            ctx.render_into(W, H, acc, per, depth, seed=0xC0FFEE, spp_begin=i * per, flags=api.RENDER_CONTINUE if i else 0)
        torch.cuda.synchronize(dev)
        dt16 = time.perf_counter() - t0
        same = bool(torch.equal(acc.view(torch.int32), ref.view(torch.int32)))
        out["as_16_passes_of_64spp"] = {"ms_per_frame": round(1e3 * dt16, 3), "over_one_call": round(dt16 / (dt / frames), 4), "same_bits_as_one_call": same}
    return out


DEFAULT_TAIL_MB, TILE_SPLIT_TAIL_MB = 1400, 20480   # rt_set_tail_limit: the library's default; what a rank of a tile-split job (owning its GPU) allows
XGMI_LINK_GBPS = 153.0   # one xGMI link of an MI355X (7 per GPU); a ring collective is bound by one link


def shares_run(ctx, scenes, torch, dev, frames=3):
    """The one-GPU bound on the 8-GPU curve. No node with more than one GPU is available to the builder, so
    the scaling curve itself cannot be measured; what CAN be measured on one GPU is the time rank 0 of a G-rank job spends in its
    render call: for the spp-split (c3) samples [0, spp / G) of every pixel, for the tile-split (c5) the interleaved bands 0, G,
    2G, ... at full spp. With the exchange ESTIMATED (ring over one xGMI link: all-reduce 2 (G-1)/G S / B, all-gather (G-1)/G S /
    B, S = the 132.7 MB accumulator, B = 153 GB/s) the efficiency of G ranks cannot exceed
    t_full / (G (t_share + t_exchange_est)): a CEILING (load imbalance between ranks and RCCL's real time come on top)."""
    out = {}
    for name, kind in (("c3", "spp"), ("c5", "tile")):
        wl = WORKLOADS[name]
        W, H, spp, depth = wl["W"], wl["H"], wl["spp"], wl["depth"]
        ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        bytes_acc = W * H * 16
        # a rank of a tile-split job owns its GPU: the colour buffer of the frame's end may hold the whole share (768 B per tile and
        # sample: 12.2 GB for the 1/8 share at 1024 spp), which makes the share parallel over samples too (rt_set_tail_limit)
        ctx.set_tail_limit(TILE_SPLIT_TAIL_MB if kind == "tile" else DEFAULT_TAIL_MB)

        def timed(G, n):
            kw = dict(spp_begin=0) if kind == "spp" else dict(band_first=0, band_stride=G)
            s = spp // G if kind == "spp" else spp
            # warm-up of this shape (c3: the same samples, so that the colour buffer is there; c5 keeps its buffer from share to share
            # once the first two calls have grown it, and its G = 8 share fits what they left)
            ctx.render_into(W, H, acc, s if name == "c3" else min(s, 8), depth, seed=0xC0FFEE, **kw)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            k_ms = 0.0
            for _ in range(n):
                ctx.render_into(W, H, acc, s, depth, seed=0xC0FFEE, **kw)
                k_ms += ctx.last_kernel_ms()[0]
            torch.cuda.synchronize(dev)
            return 1e3 * (time.perf_counter() - t0) / n, k_ms / n

        n_full = 1 if name == "c5" else frames
        t_full, k_full = timed(1, n_full)
        rows = {}
        for G in (2, 4, 8):
            t, k = timed(G, frames if name == "c3" else 1)
            ex = ((2.0 if kind == "spp" else 1.0) * (G - 1) / G * bytes_acc / (XGMI_LINK_GBPS * 1e9)) * 1e3
            rows[str(G)] = {"t_share_ms": round(t, 3), "kernel_ms": round(k, 3), "G_t_share_over_t_full": round(G * t / t_full, 4),
                            "t_exchange_est_ms": round(ex, 3), "efficiency_ceiling": round(t_full / (G * (t + ex)), 4)}
        ctx.set_tail_limit(DEFAULT_TAIL_MB)
        out[name] = {"workload": wl["desc"], "split": kind, "tail_limit_mb": TILE_SPLIT_TAIL_MB if kind == "tile" else DEFAULT_TAIL_MB, "t_full_ms": round(t_full, 3), "kernel_full_ms": round(k_full, 3), "ranks": rows}
        del acc
    out["note"] = ("one GPU: the render call of rank 0 of a G-rank job (c3: samples [0, spp/G) of every pixel; c5: bands 0, G, 2G, ... at "
                   "full spp), wall clock per call incl. launch and sync; exchange ESTIMATED as a ring over one 153 GB/s xGMI link on the "
                   "132.7 MB accumulator (all-reduce 2(G-1)/G S/B, all-gather (G-1)/G S/B); efficiency_ceiling = t_full / (G (t_share + "
                   "t_exchange_est)) is an upper bound, NOT a measured scaling curve")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)   # (a C3 frame varies by +-1 % from frame to frame: which wave gets which tile at the end)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (diagnostics only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the short runs of the other workloads (the `also` dict)")
    ap.add_argument("--no-shares", action="store_true", help="skip the per-rank share timings (also.shares)")
    ap.add_argument("--ordered-reduce", action="store_true", help="N > 1, spp-split: sum the accumulators in rank order (all-gather; bit-reproducible) instead of one all-reduce")
    ap.add_argument("--shading-sort", type=int, default=-1, help="rt_set_shading_sort mode (diagnostics only; default: the library's)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ray_amd import api, dist as rdist, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    # (RCCL refuses two ranks on one GPU). Never set by the driver; numbers from such a run mean nothing.
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # communicator set-up (RCCL creates its rings on the first collective) is not part of a frame
        dist.all_reduce(torch.zeros(1, dtype=torch.float32, device=dev))

    wl = dict(WORKLOADS[args.workload])
    if args.spp:
        wl["spp"] = args.spp
    W, H, spp, depth = wl["W"], wl["H"], wl["spp"], wl["depth"]
    scene = scenes.get_scene(wl["scene"], **wl["kw"])
    flat = scene.flatten()
    fpr = flops_per_ray(flat)

    ctx = api.Context(local_rank)
    if args.shading_sort >= 0:
        ctx.set_shading_sort(args.shading_sort)
    ctx.upload_scene(scene)
    # the library's kernels go on torch's current stream: ordered after the previous step's collective and
    # any fill of the accumulator, and the collective of this step is ordered after the kernel
    ctx.set_stream(torch.cuda.current_stream(dev))
    split = wl["split"] if world > 1 else "none"
    if split == "tile":
        ctx.set_tail_limit(TILE_SPLIT_TAIL_MB)
    # work split (ray_amd/dist.py): spp-split = every rank renders all pixels and a slice of the samples, then ONE
    # all-reduce of the accumulators; tile-split = every rank renders all samples of its interleaved 8-row bands,
    # then ONE all-gather of the disjoint bands (no zero fill, no sum)
    pl = rdist.plan(wl["split"], W, H, spp, rank, world)
    rows = rdist.padded_rows(H, world) if split == "tile" else H
    accum = torch.zeros((rows, W, 4), dtype=torch.float32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    s0, s1 = pl["spp_begin"], pl["spp_begin"] + pl["spp_count"]
    y0, yc = pl["y_begin"], pl["y_count"]
    seed = 0xC0FFEE

    def step():
        if pl["zero_first"]:
            accum.zero_()
        ctx.render_into(W, H, accum, s1 - s0, depth, seed=seed, flags=0, y_begin=y0, y_count=yc, spp_begin=s0, rays=rays,
                        band_first=pl["band_first"], band_stride=pl["band_stride"])
        ms, _ = ctx.last_kernel_ms()
        if split == "tile":
            rdist.gather_bands(accum, rank, world)
        else:
            rdist.reduce_accumulators(accum, world, ordered=args.ordered_reduce)
        return ms

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    rays.zero_()
    sync()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        kernel_ms += step()
    sync()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    rr = rays.clone()
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr)
    elapsed, kernel_ms = float(tt[0]), float(tt[1])
    total_rays = int(rr.item())
    local_rays = int(rays.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        mrays = total_rays / elapsed / 1e6
        # roofline of the dominant kernel (render_kernel) on this rank: algorithmic flops of the
        # closest-hit tests it executed / its average launch duration (HIP events, library stream)
        k_ms = kernel_ms / max(args.steps, 1)
        flops_launch = local_rays / max(args.steps, 1) * fpr
        achieved = flops_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic = None   # HBM bytes per launch from the committed PMC passes, only for the exact configuration they ran
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(args.workload)
            if tr and world == 1 and (tr["W"], tr["H"], tr["spp"]) == (W, H, spp):
                traffic = tr["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "Mrays/s (incl. bounces)", "value": round(mrays, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["desc"], "name": args.workload, "width": W, "height": H, "spp": spp, "depth": depth,
                       "primitives": int(flat["n_prims"]), "split": wl["split"] if world > 1 else "none",
                       "rays_per_frame": total_rays // max(args.steps, 1), "seed": seed},
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved, 4), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 5), "traffic": traffic,
                         "kernel": "render_kernel (kernel_ms includes tail_sum_kernel, which adds the colours of the frame's last part in "
                                   "sample order: 0.1 % of the time, DESIGN.md 4.5b)",
                         "kernel_ms": round(k_ms, 3), "flops_per_ray": fpr,
                         "traffic_note": "the accumulator is 16 B per pixel; the rest of `traffic` is the colour buffer of the frame's end "
                                         "(bounded by rt_set_tail_limit, default 1400 MiB), written and read once, and the HBM part of the path "
                                         "stack beyond six bounces (profiles/traffic.json): < 1 % of HBM bandwidth",
                         "accounting": "algorithmic (SURVEY 8d): closest-hit scans x primitives x flops of the reference's test; "
                                       "the box hierarchy and the packed filter answer most tests without evaluating them",
                         "accumulator_GBps": round(W * H * 16 / (k_ms * 1e-3) / 1e9, 3) if k_ms > 0 else None},
        }
        try:   # executed work from the committed PMC passes, only for the exact configuration they ran
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc.json"))).get(args.workload)
            if pmc and world == 1 and (pmc["W"], pmc["H"], pmc["spp"]) == (W, H, spp):
                ex = executed_work(pmc, k_ms)
                ex["valu_insts_per_ray"] = round(pmc["counters_per_launch"]["SQ_INSTS_VALU"] * 64 / max(local_rays / max(args.steps, 1), 1), 1)
                out["roofline"]["executed"] = ex
        except (OSError, ValueError, KeyError) as e:
            out["roofline"]["executed"] = {"error": str(e)[:100]}
        if world == 1 and not args.no_also:
            # the other single-GPU workloads, one timed frame each, so that 1080p (BASELINE.json configs[1]) and the triangle
            # configs are driver-observed as well; their rocprof summaries are profiles/r3_<name>_*
            out["also"] = {n: quick_run(ctx, api, scenes, torch, dev, n) for n in ("c2", "box", "c4", "c5") if n != args.workload}
            if not args.no_shares:
                out["also"]["shares"] = shares_run(ctx, scenes, torch, dev)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
