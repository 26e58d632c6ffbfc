#!/usr/bin/env python3
"""Frames/s through the C++ host classes the way the reference's MainController drives them: HybKinectfu::processNewFrame per
frame (host depth image in, blocking pose read-back out), C2 geometry (GPU).  Compare: bench.py keeps frames and poses on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import host_app as H, scene as S
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
wl = bench.workload(1, "c2")
cam = wl["cam"]
frames, _ = S.make_stream(100, cam, wl["size"])
frames = [np.ascontiguousarray(f) for f in frames]
for host_loop in (False, True):
    app = H.App(wl["res"], wl["size"], cam, host_loop=host_loop)
    for k in range(10):
        app.process_frame(frames[k], k)
    t0 = time.perf_counter()
    lost = 0
    for k in range(10, 10 + steps):
        lost += 0 if app.process_frame(frames[k % 100], k) else 1
    dt = time.perf_counter() - t0
    print("processNewFrame, %s: %.1f frames/s (%.3f ms per frame), lost %d" % (
        "reference's host Gauss-Newton loop (19 read-backs per frame)" if host_loop else "device-resident tracking loop, one read-back per frame",
        steps / dt, 1e3 * dt / steps, lost))
    app.close()
