#!/usr/bin/env python3
"""How selective is the brick cull?  Bricks queued vs voxels actually updated per frame (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
ctx = K.Context(K.camera(*cam), wl["res"], wl["size"], P["volume_max_weight"], levels=3)
for k in range(4):
    pose = S.trajectory_pose(k, wl["size"]).astype(np.float32)
    ctx.upload_depth_mm(S.render_depth_mm(pose, cam, wl["size"]))
    ctx.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
    s = ctx.stats()
    print("%s frame %d: bricks queued %d of %d, voxels updated %d = %.1f per queued brick (512 max)" % (
        cfg, k, s["bricks_active"], s["bricks_total"], s["updated_last"], s["updated_last"] / max(s["bricks_active"], 1)))
