#!/usr/bin/env python3
"""Tracking quality over a long run of the C2 / C4 stream: distance of the tracked pose from the ground-truth trajectory every few
hundred frames, lost frames.  usage: track_drift.py [c2|c4] [frames]   (compare builds with KF_LIB=...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
wl = bench.workload(1, cfg)
cam = wl["cam"]
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
worst_t = worst_r = 0.0
for k in range(n):
    pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
    if k % 97 == 5 or k == n - 1:
        pipe.sync()
        ok, pose, status, iters = pipe.ctx.track_result()
        gt = S.trajectory_pose(k, wl["size"])
        dt = float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]))
        dr = float(np.arccos(np.clip((np.trace(pose[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
print("%s %d frames: lost %d, worst |t - gt| = %.2e m, worst rotation error = %.2e rad (sampled every 97 frames), last pose bits %s" % (
    cfg, n, pipe.stats()["frames_lost"], worst_t, worst_r, pose.astype(np.float32).tobytes().hex()[:24]))
