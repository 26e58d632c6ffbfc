#!/usr/bin/env python3
"""Why is the first host-fed context of a process slower (BENCH_r04: pcie_inclusive.runs 2945 / 4912 / 4812)?
Runs, each in a FRESH process: (a) 4 contexts, warm-up 10 frames each; (b) the same with a warm-up of 150 frames for the first context only;
(c) a resident-stream context first (what bench.py runs before this leg), then (a).  Prints the per-context rates."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode):
    import numpy as np, torch
    import bench
    from hybkinectfu_amd import lib as K, scene as S
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    bench.K = K
    wl = bench.workload(1, "c2")
    cam = wl["cam"]
    frames, _ = S.make_stream(100, cam, wl["size"])
    host = [np.ascontiguousarray(f, np.uint16) for f in frames]
    frame_of = lambda k: host[k % len(host)]
    if mode == "c":
        dev = torch.from_numpy(frames.astype(np.int16)).cuda()
        fb = cam[0] * cam[1] * 2
        p = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl)
        for k in range(200):
            p.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
        p.sync(); p.close()
    out = []
    for r in range(4):
        warm = 150 if (mode == "b" and r == 0) else 10
        pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl)
        t_w = time.perf_counter()
        for k in range(warm):
            pipe.process_frame_host(frame_of, k)
        pipe.sync()
        t0 = time.perf_counter()
        seg = []
        for k in range(warm, warm + 100):
            pipe.process_frame_host(frame_of, k)
            if (k - warm) % 25 == 24:
                pipe.sync(); seg.append(time.perf_counter())
        pipe.sync()
        dt = time.perf_counter() - t0
        quarters = [round(25 / (b - a), 0) for a, b in zip([t0] + seg[:-1], seg)]
        out.append("%.0f (warm-up %d frames took %.1f ms; quarters %s)" % (100 / dt, warm, 1e3 * (t0 - t_w), quarters))
        pipe.close()
    print("mode %s: " % mode + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for m in ("a", "b", "c", "a"):
            subprocess.run([sys.executable, os.path.abspath(__file__), m], check=False)
