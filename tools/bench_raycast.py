#!/usr/bin/env python3
"""Micro-benchmark of the raycast pass alone on C2 / C4 after fusing a few frames (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
ctx = K.Context(K.camera(*cam), wl["res"], wl["size"], P["volume_max_weight"], levels=3)
for k in range(4):
    pose = S.trajectory_pose(k, wl["size"]).astype(np.float32)
    ctx.upload_depth_mm(S.render_depth_mm(pose, cam, wl["size"]))
    ctx.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
ctx.stage_timers(1 << 4)
for _ in range(30):
    ctx.raycast(pose, 0.035, P["depth_trunc_min"], wl["trunc_max"])
ms, cnt = ctx.read_stage_ms()
v = ctx.download_map(K.MAP_MODEL_VERTICES)
print("%s raycast: %.4f ms per frame, %d hit pixels of %d" % (cfg, ms[4] / cnt[4], int((v[..., 3] != 0).sum()), v.shape[0] * v.shape[1]))
if os.environ.get("KF_RAYCAST_EXP") == "3":
    n = ctx.download_map(K.MAP_MODEL_NORMALS)
    for i, name in enumerate(["prologue ticks", "march ticks", "eval ticks", "loop trips"]):
        a = v[..., i]
        print("%-15s mean %9.0f  p50 %9.0f  p99 %9.0f  max %9.0f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 99), a.max()))
    a = n[..., 0]
    print("%-15s mean %9.1f  p50 %9.0f  p99 %9.0f  max %9.0f" % ("voxel samples", a.mean(), np.percentile(a, 50), np.percentile(a, 99), a.max()))
    # per 8x8 patch (= one wave): the wave runs as long as its slowest lane
    trips = v[..., 3].reshape(v.shape[0] // 8, 8, v.shape[1] // 8, 8).max(axis=(1, 3))
    print("per-wave max trips: mean %.1f  p99 %.0f  max %.0f" % (trips.mean(), np.percentile(trips, 99), trips.max()))
