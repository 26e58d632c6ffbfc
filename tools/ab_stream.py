#!/usr/bin/env python3
"""One line per run: frames/s and per-stage device times of a BASELINE configuration's stream on one GPU -- for A/B runs of library variants /
environment switches inside ONE gpurun call (boxes differ by up to 15 %).  usage: ab_stream.py [c2|c4|c5|c1|c3] [timed frames] [frames fused before] [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
bench.K = K
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 10
label = sys.argv[4] if len(sys.argv) > 4 else ""
wl = bench.workload(1, cfg)
cam = wl["cam"]
nu = 100 if cfg != "c5" else 12
frames, _ = S.make_stream(nu, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
at = (lambda k: k % nu) if nu == 100 else (lambda k: bench.ping_pong(k, nu))
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
def run(a, b):
    for k in range(a, b):
        pipe.process_frame_device(dev.data_ptr() + at(k) * fb, k, dev.data_ptr() + at(k + 1) * fb)
run(0, pre); pipe.sync()
s0 = pipe.stats()
t0 = time.perf_counter()
run(pre, pre + n); pipe.sync()
dt = time.perf_counter() - t0
s1 = pipe.stats()
pipe.stage_timers(0x1F | (1 << 5) | (1 << 7))
run(pre + n, pre + n + 40); pipe.sync()
ms, cnt = pipe.read_stage_ms()
us = lambda i: 1000.0 * float(ms[i]) / max(int(cnt[i]), 1)
print("%-28s %s frames %d..%d: %8.1f frames/s | track %6.1f integrate %6.1f (kernel %6.1f) raycast %6.1f (kernel %6.1f) us | queued %d lost %d n_upd %d" % (
    label or str(bench.env_knobs()), cfg, pre, pre + n, n / dt, us(2), us(3), us(5), us(4), us(7), s1["bricks_active"], s1["frames_lost"] - s0["frames_lost"],
    (s1["updated_total"] - s0["updated_total"]) // n))
