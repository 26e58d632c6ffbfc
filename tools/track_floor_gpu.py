#!/usr/bin/env python3
"""GPU: pose divergence of the HIP path from the oracle's sequence (tests/golden/floor_*.npz) over the same frames, per frame.
usage: track_floor_gpu.py [c2|h128]   (KF_LIB=... selects a library variant; KF_ICP_PERSISTENT=0 the per-step launch form)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
P = S.STOCK
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
g = np.load(os.path.join(ROOT, "tests", "golden", "floor_%s.npz" % name))
meta, o_poses = json.loads(str(g["meta"])), g["poses"]
cam, res, size, kw = tuple(meta["cam"]), meta["res"], meta["size"], meta["kw"]
cam = (int(cam[0]), int(cam[1])) + tuple(cam[2:])
trunc, dist = kw.get("sdf_trunc", P["integrate_sdf_trunc"]), kw.get("integ_dist", P["integrate_depth_trunc"])
ctx = K.Context(K.camera(*cam), res, size, P["volume_max_weight"], levels=3)
ctx.set_pose(S.pose0(size))
worst_t = worst_r = 0.0
per = []
for k in range(len(o_poses)):
    ctx.upload_depth_mm(S.render_depth_mm(S.trajectory_pose(k, size), cam, size))
    ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    ctx.integrate(None, trunc, dist)
    ctx.raycast(None, P["raycast_increment_factor"] * trunc, P["depth_trunc_min"], P["depth_trunc_max"])
    ok, pose, status, iters = ctx.track_result()
    dt = float(np.max(np.abs(pose[:3, 3].astype(np.float64) - o_poses[k][:3, 3])))
    dr = float(np.max(np.abs(pose[:3, :3].astype(np.float64) - o_poses[k][:3, :3])))
    per.append((dt, dr)); worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
    assert ok, (k, status)
print(json.dumps(dict(case=name, lib=os.environ.get("KF_LIB", "product"), persistent=os.environ.get("KF_ICP_PERSISTENT", "1"),
                      frames=len(o_poses), gpu_vs_oracle_dt_m=worst_t, gpu_vs_oracle_dr=worst_r, floor_dt_m=meta["floor_dt_m"], floor_dr=meta["floor_dr"],
                      ratio_t=round(worst_t / meta["floor_dt_m"], 2), ratio_r=round(worst_r / meta["floor_dr"], 2),
                      per_frame_dt=[round(p[0], 9) for p in per])))
