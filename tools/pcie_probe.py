#!/usr/bin/env python3
"""Diagnostic: host-inclusive times of one frame (scene upload, render, tonemap + image to host, accumulators to host).
    python tools/pcie_probe.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_amd import api, scenes
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
sc = scenes.get_scene(wl['scene'], **wl['kw'])
ctx = api.Context(0)
W, H = wl['W'], wl['H']
acc = ctx.alloc(W * H * 16).zero()
for rep in range(3):
    t0 = time.perf_counter(); ctx.upload_scene(sc); t1 = time.perf_counter()
    ctx.render_into(W, H, acc, wl['spp'], wl['depth'], seed=1); t2 = time.perf_counter()
    img = ctx.tonemap(W, H, acc); t3 = time.perf_counter()
    a = acc.to_numpy('float32', (H, W, 4)); t4 = time.perf_counter()
    print(f"upload_scene {1e3*(t1-t0):.2f} ms, render {1e3*(t2-t1):.1f} ms, tonemap + RGB32 image to host {1e3*(t3-t2):.2f} ms, float accumulators to host {1e3*(t4-t3):.2f} ms")
