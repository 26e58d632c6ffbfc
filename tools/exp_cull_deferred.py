#!/usr/bin/env python3
"""Experiments build only (KF_LIB=.../libhybkf_exp.so KF_INTEGRATE_EXP=16): the bricks the cull QUEUES although quarters of them are in a deferred
state -- how many, and which of the whole-brick retirement's conditions they fail.  usage: exp_cull_deferred.py [c4|c5|c2] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
wl = bench.workload(1, cfg)
cam = wl["cam"]
nf = 100 if cfg != "c5" else 12
frames, _ = S.make_stream(nf, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
from hybkinectfu_amd.pipeline import SingleGpuPipeline
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
c = pipe.ctx
for lo, hi in ((0, 2), (2, 10), (10, n)):
    a0, b0, m0, _ = c.work_counters()
    for k in range(lo, hi):
        pipe.process_frame_device(dev.data_ptr() + (k % nf) * fb, k)
    pipe.sync()
    a, b, m, _ = c.work_counters()
    a, b, m = a - a0, b - b0, m - m0
    f = float(hi - lo)
    lo32, hi32 = (lambda x: (x & 0xFFFFFFFF) / f), (lambda x: (x >> 32) / f)
    if os.environ.get("KF_INTEGRATE_EXP") == "17":
        print("%s frames %3d..%3d: queued %7d bricks (last frame) | per frame: kept by the cull %7.0f | an EXACT footprint would drop %6.0f and retire %6.0f | a 4-pixel table level would drop %6.0f and retire %6.0f" % (
            cfg, lo, hi, c.stats()["bricks_active"], lo32(a), hi32(a), lo32(b), hi32(b), lo32(m)))
        continue
    print("%s frames %3d..%3d: queued %7d bricks (last frame) | per frame: all four quarters deferred yet queued %7.0f = no tile test %6.0f + may leave the image %6.0f + "
          "a pixel without depth %6.0f + surface within a truncation distance of the tiles' minimum %6.0f | one to three quarters deferred %7.0f" % (
              cfg, lo, hi, c.stats()["bricks_active"], lo32(a), hi32(a), lo32(b), hi32(b), lo32(m), hi32(m)))
