#!/usr/bin/env python3
"""C2 geometry with the reference's stock switches use_color=1, color_angle_weight=1 (src/config.ini): the colour planes ride along
in integrate and raycast.  One colour image is uploaded once and reused; depth frames are resident (GPU)."""
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
inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
for color in (False, True):
    c = K.Context(K.camera(*cam), wl["res"], wl["size"], P["volume_max_weight"], levels=3, has_color=color)
    c.set_pose(S.pose0(wl["size"]))
    if color:
        c.upload_rgb(np.random.default_rng(1).integers(0, 256, (cam[1], cam[0], 3)).astype(np.uint8))
    def frame(k):
        c.set_depth_mm_device(dev.data_ptr() + (k % 100) * fb)
        c.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        c.integrate(None, P["integrate_sdf_trunc"], wl["integ_dist"], has_color=color, angle_weight=color)
        c.raycast(None, inc, P["depth_trunc_min"], wl["trunc_max"], has_color=color)
    for k in range(10):
        frame(k)
    c.sync()
    c.stage_timers(0x18)
    t0 = time.perf_counter()
    for k in range(10, 10 + steps):
        frame(k)
    c.sync()
    dt = time.perf_counter() - t0
    ms, cnt = c.read_stage_ms()
    print("colour %s: %.1f frames/s (%.3f ms per frame; integrate %.3f ms, raycast %.3f ms), lost %d" % (
        "on " if color else "off", steps / dt, 1e3 * dt / steps, ms[3] / cnt[3], ms[4] / cnt[4], c.stats()["frames_lost"]))
    c.close()
