#!/usr/bin/env python3
"""Micro-benchmark of the TSDF fusion pass alone: one frame integrated repeatedly with a fixed pose (GPU).
usage: tools/bench_integrate.py [c2|c4] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
ctx = K.Context(K.camera(*cam), wl["res"], wl["size"], P["volume_max_weight"], levels=3)
pose = S.trajectory_pose(0, wl["size"]).astype(np.float32)
ctx.upload_depth_mm(S.render_depth_mm(pose, cam, wl["size"]))
ctx.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
for _ in range(3):
    ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
ctx.sync()
st = ctx.stats()
ctx.stage_timers((1 << 5) | (1 << 3))
for _ in range(reps):
    ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
ms, cnt = ctx.read_stage_ms()
k = ms[5] / cnt[5]
alg = st["updated_last"] * 16 + cam[0] * cam[1] * 4
print("%s: N_upd %d, active bricks %d / %d, fusion kernel %.4f ms (all passes %.4f ms), algorithmic %.1f MB -> %.0f GB/s (%.1f%% of 8 TB/s)"
      % (cfg, st["updated_last"], st["bricks_active"], st["bricks_total"], k, ms[3] / cnt[3], alg / 1e6, alg / k / 1e6, alg / k / 1e6 / 80))
