import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SingleGpuPipeline
res, size, cam = 2048, 8.0, S.vga_camera(2)
wl = dict(trunc_max=8.0, integ_dist=8.0)
n = 30
frames, _ = S.make_stream(n, cam, size)
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
pipe = SingleGpuPipeline(K.camera(*cam), res, size, wl)
for k in range(8):
    pipe.process_frame_device(dev.data_ptr() + k * fb, k)
pipe.sync()
pipe.stage_timers(0xFF)
t0 = time.perf_counter()
for k in range(8, n):
    pipe.process_frame_device(dev.data_ptr() + k * fb, k)
pipe.sync()
dt = (time.perf_counter() - t0) / (n - 8)
ms, cnt = pipe.read_stage_ms()
names = ["upload", "preprocess", "track", "integrate", "raycast", "integrate_kernel", "mcubes", "raycast_kernel"]
print("C5 on one GPU (2048^3 @ 8 m, 1280x960): %.3f ms/frame = %.0f frames/s;" % (dt * 1e3, 1 / dt), ", ".join("%s=%.3f" % (nm, ms[i] / max(int(cnt[i]), 1)) for i, nm in enumerate(names)), "lost", pipe.stats()["frames_lost"])
