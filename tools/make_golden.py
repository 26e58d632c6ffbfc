#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/libkforacle.so).

The reference ships no golden vectors for this path, and its kernels cannot be built here
(nvcc absent), so these fixtures are produced by the restatement, whose helper arithmetic is pinned against the
reference's own headers (tests/test_oracle_vs_ref.py) and whose end-to-end counts reproduce the reference run recorded in
SURVEY.md section 8c (tests/test_oracle_golden.py::test_reference_recorded_smoke_numbers).  They freeze the oracle's
behaviour so that later edits to oracle/ cannot drift silently.   usage: python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from hybkinectfu_amd import scene as S  # noqa: E402

P = S.STOCK


def case(name, res, size, cam, trunc, n_frames):
    ocam = O.Cam.make(*cam)
    vol = O.OVolume(res, size, P["volume_max_weight"])
    out = {}
    n_upd = []
    for k in range(n_frames):
        pose = S.trajectory_pose(3 * k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        d = O.depth_mm_to_m(mm)
        tr = O.trunc_depth(d, P["depth_trunc_min"], P["depth_trunc_max"])
        fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        v = O.depth_to_vertices(fl, ocam)
        n = O.vertices_to_normals(v)
        n_upd.append(O.integrate(vol, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam))
        out["mm%d" % k] = mm
        out["pose%d" % k] = pose
    out["filtered_last"] = fl
    out["vertices_last"] = v
    out["normals_last"] = n
    out["v_l2"] = O.pyramid(v, 3)[2]
    out["n_l2"] = O.pyramid(n, 3, normals=True)[2]
    out["n_upd"] = np.array(n_upd, np.int64)
    out["tsdf"] = vol.tsdf.copy()
    out["weight"] = vol.weight.copy()
    mv, mn, _ = O.raycast(vol, False, pose, 0.7 * trunc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    out["model_v"], out["model_n"] = mv, mn
    tris = O.marching_cubes(vol, False, 300 * size / res, 400000)
    out["tri_pos"] = tris["v"]["pos"].copy()
    nxt = S.trajectory_pose(3 * n_frames, size).astype(np.float32)
    mm = S.render_depth_mm(nxt, cam, size)
    tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
    fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
    v = O.depth_to_vertices(fl, ocam)
    n = O.vertices_to_normals(v)
    sd, sf, valid = O.icp_system(v, n, mv, mn, ocam, pose, O.mat44_inverse(pose), P["icp_thre_dist"], P["icp_thre_sin_angle"])
    out["mm_next"] = mm
    out["icp27"] = sd
    out["icp_valid"] = np.array([valid])
    ok, p1 = O.icp_estimate(O.pyramid(v, 3), O.pyramid(n, 3, True), O.pyramid(mv, 3), O.pyramid(mn, 3, True), ocam,
                            P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    out["icp_ok"] = np.array([int(ok)])
    out["icp_pose"] = p1
    sd2, _, valid2 = O.sdf_system(vol, tr, ocam, pose)
    out["sdf27"] = sd2
    out["sdf_valid"] = np.array([valid2])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(name, "n_upd", n_upd, "tris", len(tris), "icp_valid", valid, "sdf_valid", valid2, "icp_ok", ok)


if __name__ == "__main__":
    case("s32", 32, 3.0, (64, 48, 31.5, 23.5, 52.5, 52.5), 5 * 3.0 / 32, 2)
    case("s64", 64, 3.0, (160, 120, 79.5, 59.5, 131.25, 131.25), 5 * 3.0 / 64, 2)
