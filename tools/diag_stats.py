#!/usr/bin/env python3
"""Print volume/cull statistics for the bench workload after a few frames (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
import bench
wl = bench.workload(1)
cam = wl["cam"]
frames, _ = S.make_stream(10, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl)
for k in range(10):
    pipe.process_frame_device(dev.data_ptr() + k * cam[0] * cam[1] * 2, k)
    pipe.sync()
    print(k, pipe.stats(), pipe.track_result()[0])
