import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, oracle_lib as O
from hybkinectfu_amd import scene as S
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
cam = S.vga_camera(); ocam = O.Cam.make(*cam); size, res = 4.0, 512
mm = S.render_depth_mm(S.trajectory_pose(0, size), cam, size)
tr = O.trunc_depth(O.depth_mm_to_m(mm), 0.3, 4.0); n = O.vertices_to_normals(O.depth_to_vertices(tr, ocam))
for th in (8, 16, 32, 64, 128, 256):
    O.set_threads(th)
    vol = O.OVolume(res, size, 128.0)
    t0 = time.perf_counter(); O.integrate(vol, tr, n, None, False, False, S.pose0(size), 0.05, 2.0, ocam, ocam); t1 = time.perf_counter()
    O.raycast(vol, False, S.pose0(size), 0.035, ocam, 0.3, 4.0); t2 = time.perf_counter()
    print(th, "threads: integrate %.3f s raycast %.3f s" % (t1 - t0, t2 - t1))
