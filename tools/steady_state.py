#!/usr/bin/env python3
"""Frames/s and fusion-kernel time of the C2 / C4 stream early (weights still growing) and late (after max_weight = 128 frames of
the periodic stream, when most of the free space the pass touches no longer changes).  usage: steady_state.py [c2|c4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
from hybkinectfu_amd.pipeline import SingleGpuPipeline
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
c = pipe.ctx
def run(lo, hi, timers):
    c.stage_timers(timers)
    pipe.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(lo, hi):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
    pipe.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, cnt = c.read_stage_ms()
    return (hi - lo) / dt, (ms[5] / cnt[5] if cnt[5] else 0.0)
run(0, 10, 0)
for lo, hi in ((10, 60), (60, 110), (200, 250), (250, 300), (300, 400)):
    if lo == 200: run(110, 200, 0)
    fps, _ = run(lo, hi, 0)
    print("%s frames %3d..%3d: %7.1f frames/s" % (cfg, lo, hi, fps))
c.stage_timers((1 << 5) | (4 << 8))
for k in range(400, 480):
    pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
pipe.sync()
ms, cnt = c.read_stage_ms()
print("%s fusion kernel in the steady state: %.1f us (HIP events, every 4th frame)" % (cfg, 1e3 * ms[5] / max(cnt[5], 1)), "lost", c.stats()["frames_lost"])
