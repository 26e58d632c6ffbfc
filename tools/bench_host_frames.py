#!/usr/bin/env python3
"""Frames/s when every depth frame comes from HOST memory (kf_upload_depth_mm) instead of lying in HBM: the PCIe-inclusive rate
that DESIGN.md quotes beside the headline number (GPU).  usage: tools/bench_host_frames.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
wl = bench.workload(1, "c2")
cam, P = wl["cam"], S.STOCK
frames, _ = S.make_stream(100, cam, wl["size"])
frames = [np.ascontiguousarray(f) for f in frames]
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl)
c = pipe.ctx
def frame(k):
    c.upload_depth_mm(frames[k % 100])
    c.preprocess(P["depth_trunc_min"], pipe.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
    c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    c.integrate(None, P["integrate_sdf_trunc"], pipe.integ_dist)
    c.raycast(None, pipe.inc, P["depth_trunc_min"], pipe.trunc_max)
for k in range(10):
    frame(k)
pipe.sync()
t0 = time.perf_counter()
for k in range(10, 10 + steps):
    frame(k)
pipe.sync()
dt = time.perf_counter() - t0
print("host-fed frames (614 KB over PCIe each): %.1f frames/s, %.3f ms per frame; lost %d" % (steps / dt, 1e3 * dt / steps, pipe.stats()["frames_lost"]))
