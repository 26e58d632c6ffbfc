#!/usr/bin/env python3
"""Experiments build only (KF_LIB=.../libhybkf_exp.so KF_INTEGRATE_EXP=13): what the fusion pass's waves (one per quarter brick) do per frame on
the C2 / C4 / C5 stream -- skipped (saturated / deferred), nothing to update, flush + write, written (band / whole free space / partial free
space) -- and how many bricks the cull queues.  usage: exp_wave_kinds.py [c2|c4|c5] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
nf = 100 if cfg != "c5" else 12
frames, _ = S.make_stream(nf, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
from hybkinectfu_amd.pipeline import SingleGpuPipeline
pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
c = pipe.ctx
edges = [0, 1, 2, 3, 5, 10, 20, 40, 80, 130, 200, 400]
for lo, hi in zip(edges, edges[1:]):
    if lo >= n:
        break
    hi = min(hi, n)
    a0, b0, m0, _ = c.work_counters()          # (the counters are never reset here: kf_stage_timers(1 << 16) would also let the raycast count into them)
    for k in range(lo, hi):
        pipe.process_frame_device(dev.data_ptr() + (k % nf) * fb, k)
    pipe.sync()
    a, b, m, _ = c.work_counters()
    a, b, m = a - a0, b - b0, m - m0
    f = float(hi - lo)
    st = c.stats()
    lo32, hi32 = (lambda x: (x & 0xFFFFFFFF) / f), (lambda x: (x >> 32) / f)      # (differences of packed counts: each half stays below 2^32 over a run)
    print("%s frames %3d..%3d: queued %7d bricks (last frame) | waves/frame: skipped sat %8.0f def %8.0f | flush %7.0f | written band %8.0f whole-free %8.0f partial-free %8.0f" % (
        cfg, lo, hi, st["bricks_active"], lo32(a), hi32(a), lo32(b), hi32(b), lo32(m), hi32(m)))
