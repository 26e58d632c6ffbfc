#!/usr/bin/env python3
"""Is the frame loop GPU-bound?  Wall time the host needs to ENQUEUE n frames (ctypes calls of pipeline.SingleGpuPipeline, no synchronisation)
against the time until the device has finished them.  usage: enqueue_time.py [c2|c4] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.workload(1, cfg)
cam = wl["cam"]
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl)
for k in range(10):
    pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
pipe.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(10 + rep * n, 10 + (rep + 1) * n):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    t1 = time.perf_counter()
    pipe.sync()
    t2 = time.perf_counter()
    print("%s, %d frames: host enqueue %.1f us per frame, device done after %.1f us per frame" % (cfg, n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
