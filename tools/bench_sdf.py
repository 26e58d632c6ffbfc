#!/usr/bin/env python3
"""BASELINE.json configs[2]-like: VGA depth, 512^3 volume, CameraPoseFinderSDF (direct SDF tracking) instead of ICP, on the
synthetic stream (no TUM data on disk).  Frames resident in HBM, no host synchronisation per frame (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wl = bench.workload(1, "c2")
cam, P = wl["cam"], S.STOCK
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
c = K.Context(K.camera(*cam), wl["res"], wl["size"], P["volume_max_weight"], levels=3)
c.set_pose(S.pose0(wl["size"]))
inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
def frame(k):
    c.set_depth_mm_device(dev.data_ptr() + (k % 100) * fb)
    c.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    c.sdf_track(k, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])
    c.integrate(None, P["integrate_sdf_trunc"], wl["integ_dist"])
    c.raycast(None, inc, P["depth_trunc_min"], wl["trunc_max"])
for k in range(10):
    frame(k)
c.sync()
c.stage_timers(1 << 2)
t0 = time.perf_counter()
for k in range(10, 10 + steps):
    frame(k)
c.sync()
dt = time.perf_counter() - t0
ms, cnt = c.read_stage_ms()
ok, pose, status, iters = c.track_result()
gt = S.trajectory_pose(9 + steps, wl["size"])
print("SDF tracker, 512^3 @ 4 m, VGA: %.1f frames/s (%.3f ms per frame, tracking stage %.3f ms, last frame %d iterations), lost %d, |t - gt| = %.1e m"
      % (steps / dt, 1e3 * dt / steps, ms[2] / cnt[2], iters, c.stats()["frames_lost"], np.linalg.norm(pose[:3, 3] - gt[:3, 3])))
