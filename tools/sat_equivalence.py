#!/usr/bin/env python3
"""The fusion pass with deferred free-space weights (KF_INTEGRATE_SAT=2: from the first frame, whatever the volume size) against the plain kernels
(KF_INTEGRATE_SAT=0) on the benchmark stream at full size: same update counts and the same voxel bits after n frames.  The switch is
read once per process, so the parent runs one child per setting (one after the other; the parent itself touches no GPU) and compares
what they print.  max_weight 128 (stock) stays in the pending-count state for 127 frames, max_weight 3 saturates after three.
    python tools/sat_equivalence.py [c2|c4] [frames] [max_weight]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cfg, n, maxw):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from hybkinectfu_amd import lib as K, scene as S
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    import bench
    wl = bench.workload(1, cfg)
    wl = dict(wl, max_weight=maxw)
    cam = wl["cam"]
    frames, _ = S.make_stream(100, cam, wl["size"])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=0)
    for k in range(n):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k)
    pipe.sync()
    st = pipe.stats(observed=True)
    h = hashlib.sha256()
    res = wl["res"]
    for z0 in range(0, res, 64):                                       # the whole volume, 64 layers at a time
        t, w = pipe.ctx.download_volume(z0, z0 + 64)
        h.update(t.tobytes()); h.update(w.tobytes())
    ok, pose, status, iters = pipe.ctx.track_result()
    print("RESULT updated_total=%d weight_gt0=%d bricks_active_last=%d lost=%d pose=%s sha256=%s" % (
        st["updated_total"], st["weight_gt0"], st["bricks_active"], st["frames_lost"], pose.astype(np.float32).tobytes().hex()[:32], h.hexdigest()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
        sys.exit(0)
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n = sys.argv[2] if len(sys.argv) > 2 else "200"
    maxw = sys.argv[3] if len(sys.argv) > 3 else "128"
    out = {}
    for mode in ("0", "2"):          # 0: never defer, 2: defer whatever the volume size (the default, 1, defers from 768^3 on)
        env = dict(os.environ, KF_INTEGRATE_SAT=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg, n, maxw], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if r.returncode or not line:
            print(r.stdout, r.stderr); sys.exit(1)
        out[mode] = line[0]
        print("KF_INTEGRATE_SAT=%s %s" % (mode, line[0]))
    a, b = out["0"].split(), out["2"].split()
    same = [x for x in a if not x.startswith("bricks_active_last")] == [x for x in b if not x.startswith("bricks_active_last")]
    print("EQUIVALENT" if same else "DIFFERENT", "(%s, %s frames, max_weight %s; the queue of the last frame differs by design: retired bricks are counted, not queued)" % (cfg, n, maxw))
    sys.exit(0 if same else 2)
