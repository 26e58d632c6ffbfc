#!/usr/bin/env python3
"""Marching cubes alone: fuse a few frames of Scene S, then extract repeatedly (GPU).  usage: tools/bench_mcubes.py [c2|c4] [reps]
Touched bytes (as VERDICT r1 #7 defines them for a flag-skipping extraction): bricks whose 3x3x3 brick
neighbourhood holds a negative voxel x 4 KiB + 72 B per triangle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
res, size = wl["res"], wl["size"]
ctx = K.Context(K.camera(*cam), res, size, P["volume_max_weight"], levels=3, max_triangles=8_000_000)
for k in range(0, 12, 3):
    pose = S.trajectory_pose(k, size).astype(np.float32)
    ctx.upload_depth_mm(S.render_depth_mm(pose, cam, size))
    ctx.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
thr = 300.0 * size / res
ctx.marching_cubes(thr); ctx.clear_triangles(); ctx.sync()
ctx.stage_timers((1 << 6) | (1 << 16))
for _ in range(reps):
    ctx.clear_triangles()
    ctx.marching_cubes(thr)
ms, cnt = ctx.read_stage_ms()
_, _, bricks, tris = ctx.work_counters()
bricks //= reps
t = ms[6] / cnt[6]
touched = bricks * 4096 + tris * 72
print("%s: %d^3, %d triangles, %d of %d bricks read, extraction %.4f ms; touched %.1f MB -> %.1f GB/s (%.2f%% of 8 TB/s); dense-equivalent %.1f GB/s" % (
    cfg, res, tris, bricks, (res // 8) ** 3, t, touched / 1e6, touched / t / 1e6, touched / t / 1e6 / 80, (res ** 3 * 8 + tris * 72) / t / 1e6))
