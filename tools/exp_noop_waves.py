#!/usr/bin/env python3
"""Experiments build only (KF_LIB=.../libhybkf_exp.so KF_INTEGRATE_EXP=10): after n frames of the C2 / C4 stream, what share of the
fusion pass's waves writes back exactly the bits it read (free space whose weight has saturated)?  usage: exp_noop_waves.py [c2|c4] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
from hybkinectfu_amd.pipeline import SingleGpuPipeline
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
c = pipe.ctx
for lo, hi in ((0, 20), (20, 60), (60, 120), (120, 140), (140, 200), (200, n)):
    c.stage_timers(1 << 16)
    for k in range(lo, hi):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
    pipe.sync()
    _, _, packed, _ = c.work_counters()
    w, same = packed & 0xFFFFFFFF, packed >> 32
    print("%s frames %3d..%3d: %6.1f k waves per frame touch memory, %5.1f %% of them write back what they read" % (cfg, lo, hi, w / (hi - lo) / 1e3, 100.0 * same / max(w, 1)))
