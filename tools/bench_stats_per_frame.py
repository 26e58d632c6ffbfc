#!/usr/bin/env python3
"""A host that asks for the observed-voxel count after EVERY frame, as the reference prints it (src/cuda/integrateVolume.cu:91-94): frames/s of the C2 / C4 stream
with kf_get_volume_stats per frame.  Run with KF_OBSERVED_COUNT=0 (every call sweeps the volume: round 4's behaviour) and without (the fusion launches switch to
their COUNT instantiations after the second question; the call is then a read-back).  usage: bench_stats_per_frame.py [c2|c4] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
bench.K = K
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.workload(1, cfg)
cam = wl["cam"]
frames, _ = S.make_stream(100, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
def run(a, b, ask):
    last = None
    for k in range(a, b):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
        if ask:
            last = pipe.stats(observed=True)["weight_gt0"]
    pipe.sync()
    return last
run(0, 10, True)
t0 = time.perf_counter(); w = run(10, 10 + n, True); dt_ask = time.perf_counter() - t0
swept = pipe.ctx.count_observed_voxels()
t0 = time.perf_counter(); run(10 + n, 10 + 2 * n, False); dt_plain = time.perf_counter() - t0
print("%s, %s: kf_get_volume_stats after every frame %.1f frames/s (%.1f us per frame) | never asked %.1f frames/s (%.1f us) | count %d, swept %d %s" % (
    cfg, "KF_OBSERVED_COUNT=" + os.environ.get("KF_OBSERVED_COUNT", "auto"), n / dt_ask, 1e6 * dt_ask / n, n / dt_plain, 1e6 * dt_plain / n, w, swept, "OK" if w == swept else "MISMATCH"))
