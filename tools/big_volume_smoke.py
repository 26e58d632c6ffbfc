#!/usr/bin/env python3
"""Largest configuration of BASELINE.json on ONE GPU: 2048^3 @ 8 m (68.7 GB of voxels), 1280x960 depth, a few frames through
the whole path plus a marching-cubes extraction.  Checks that the 64-bit addressing, the brick queue, the multi-launch ICP
(1.2 M pixels do not fit the persistent loop's 256 workgroups) and the table-less raycast hold at that size (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
P = S.STOCK
res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
size, cam = 8.0, S.vga_camera(2)
wl = dict(trunc_max=8.0, integ_dist=8.0)
n = 6
frames = np.stack([S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
t0 = time.perf_counter()
pipe = SingleGpuPipeline(K.camera(*cam), res, size, wl, max_triangles=40_000_000)
pipe.sync()
print("context with %d^3 voxels created in %.1f s" % (res, time.perf_counter() - t0), flush=True)
for k in range(n):
    t0 = time.perf_counter()
    pipe.process_frame_device(dev.data_ptr() + k * fb, k)
    ok, pose, status, iters = pipe.track_result()
    dt = time.perf_counter() - t0
    gt = S.trajectory_pose(k, size)
    print("frame %d: tracked %s status %d iters %d, %.2f ms, |t - gt| = %.2e m" % (k, ok, status, iters, dt * 1e3, np.linalg.norm(pose[:3, 3] - gt[:3, 3])), flush=True)
    assert ok
st = pipe.stats()
print("stats:", st, flush=True)
t0 = time.perf_counter()
pipe.ctx.marching_cubes(300 * size / res)                 # MeshGeneratorMarchingcube.cpp: threshold 300 cells
tris = pipe.ctx.triangles()
print("marching cubes: %d triangles in %.1f ms" % (len(tris), (time.perf_counter() - t0) * 1e3), flush=True)
hit = pipe.ctx.download_map(K.MAP_MODEL_VERTICES)[..., 3] != 0
print("raycast hit pixels: %d of %d" % (hit.sum(), hit.size))
assert st["updated_last"] > 1e8 and hit.sum() > 0.5 * hit.size and len(tris) > 100000
pipe.close()
