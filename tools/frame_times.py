#!/usr/bin/env python3
"""Device time of every one of the first n frames of the C2 stream (one HIP-event pair per frame on the context's stream, frames enqueued back
to back as bench.py does): how long does the pipeline take to reach its steady state?  usage: frame_times.py [c2|c4] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
wl = bench.workload(1, cfg)
cam = wl["cam"]
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl)
pipe.stage_timers(0x1F | (1 << 5))
rows = []
for k in range(n):
    pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    pipe.sync()
    ms, cnt = pipe.read_stage_ms()
    st = pipe.stats()
    rows.append((k, [1000.0 * float(ms[i]) for i in (1, 2, 3, 4, 5)], st["updated_last"], st["bricks_active"]))
for k, us, upd, br in rows:
    print("frame %3d: preprocess %6.1f track %6.1f integrate %6.1f raycast %6.1f (fusion kernel %6.1f) us   N_upd %8d  bricks queued %6d" % (k, *us, upd, br))
