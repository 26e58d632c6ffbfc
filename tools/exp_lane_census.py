#!/usr/bin/env python3
"""Experiments build only (KF_LIB=.../libhybkf_exp.so KF_INTEGRATE_EXP=17): what the fusion pass's fetched bytes are made of (rocprofv3 reports
1.7 x the algorithmic read bytes at C2).  Per frame: queued bricks, pair lanes that load their 16 bytes, lanes of which BOTH voxels update, waves whose 64 lanes all
load (whole 1-KiB bursts) against partial ones -- and the bytes each reading implies.  usage: exp_lane_census.py [c2|c4|c1] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
wl = bench.workload(1, cfg)
cam = wl["cam"]
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
from hybkinectfu_amd.pipeline import SingleGpuPipeline
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
c = pipe.ctx
c.set_defer(0)                                  # the plain read-modify-write form (the roofline kernel)
for k in range(10):
    pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
pipe.sync()
a0, b0, m0, _ = c.work_counters()
s0 = c.stats()
for k in range(10, 10 + n):
    pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
pipe.sync()
a, b, m, _ = c.work_counters()
s1 = c.stats()
a, b, m = a - a0, b - b0, m - m0
f = float(n)
lo, hi = (lambda x: (x & 0xFFFFFFFF) / f), (lambda x: (x >> 32) / f)
n_upd = (s1["updated_total"] - s0["updated_total"]) / f
lanes, waves, both, waves_load, full, partial = lo(a), hi(a), lo(b), hi(b), lo(m), hi(m)
one = lanes - both
print("%s per frame: queued bricks %d (= %d waves, %.1f MB if every lane loaded) | N_upd %d (%.1f MB read + %.1f MB written algorithmic)" % (
    cfg, waves / 4, waves, waves * 64 * 16 / 1e6, n_upd, n_upd * 8 / 1e6, n_upd * 8 / 1e6))
print("   pair lanes that load: %d (%.1f MB = %.2f x the algorithmic read bytes) -- both voxels update %d, one voxel %d; check: 2 x both + one = %d" % (
    lanes, lanes * 16 / 1e6, lanes * 16 / max(n_upd * 8, 1), both, one, 2 * both + one))
print("   waves that load anything: %d of %d -- all 64 lanes (whole 1-KiB burst) %d, partial %d (a partial wave still moves whole 64-B sectors / 128-B lines: "
      "its masked lanes' bytes travel too)" % (waves_load, waves, full, partial))
print("   if every loading wave moved its whole 1 KiB: %.1f MB; if every loading lane's 32-B sector moved: %.1f MB" % (waves_load * 1024 / 1e6, lanes * 16 / 1e6))
pipe.close()
