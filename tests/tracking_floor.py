"""Sequence-level tracking agreement: how far may two faithful implementations of the reference's tracker drift apart?

The reference solves its 6x6 normal equations in fp32 in world coordinates; last-bit differences of the tracker's inputs (another
summation order of the 27 sums -- the reference's own tree reduction differs from any sequential order --, another `exp`
implementation in the bilateral filter, fused accumulation) are amplified along the weakly constrained direction.  This module runs the
CPU oracle's full per-frame path (preprocess -> ICP -> integrate -> raycast, src/HybKinectfu.cpp:98-160) over the first frames of a
benchmark stream, once as it is and once per perturbation, and reports the oracle-vs-oracle pose divergence: the FLOOR below which a
sequence-level comparison of two implementations cannot be held.  tests/test_tracking_floor.py (CPU) pins the floor's order of
magnitude; tests/test_gpu_tracking_floor.py asserts GPU-vs-oracle divergence <= 2 x floor on the same frames.
"""
import numpy as np

import oracle_lib as O
from hybkinectfu_amd import scene as S

P = S.STOCK
PERTURBATIONS = {"reversed_sums": 1, "exp2_taps": 2, "fused_accumulation": 4, "reciprocal_solve": 8, "all": 15}


def oracle_sequence(res, size, cam, n_frames, perturb=0, trunc_max=None, integ_dist=None, sdf_trunc=None, keep_volume=False, sdf_tracker=False):
    """poses [n,4,4] float32, tracked [n] of the oracle over frames 0..n-1 of Scene S's stream (hybkinectfu_amd/scene.py)."""
    trunc_max = P["depth_trunc_max"] if trunc_max is None else trunc_max
    integ_dist = P["integrate_depth_trunc"] if integ_dist is None else integ_dist
    sdf_trunc = P["integrate_sdf_trunc"] if sdf_trunc is None else sdf_trunc
    O.set_perturbation(perturb)
    try:
        ocam = O.Cam.make(*cam)
        vol = O.OVolume(res, size, P["volume_max_weight"])
        pose = S.pose0(size)
        poses, tracked = [], []
        mv = mn = None
        for k in range(n_frames):
            mm = S.render_depth_mm(S.trajectory_pose(k, size), cam, size)
            tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], trunc_max)
            fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
            v = O.depth_to_vertices(fl, ocam)
            n = O.vertices_to_normals(v)
            ok = True
            if k > 0 and sdf_tracker:        # CameraPoseFinderSDF (src/CameraPoseFinderSDF.cpp:44-106)
                ok, pose, _ = O.sdf_estimate(vol, tr, ocam, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
            elif k > 0:
                ok, pose = O.icp_estimate(O.pyramid(v, 3), O.pyramid(n, 3, True), O.pyramid(mv, 3), O.pyramid(mn, 3, True), ocam,
                                          P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
            if ok:
                O.integrate(vol, tr, n, None, False, False, pose, sdf_trunc, integ_dist, ocam, ocam)
            mv, mn, _ = O.raycast(vol, False, pose, P["raycast_increment_factor"] * sdf_trunc, ocam, P["depth_trunc_min"], trunc_max)
            poses.append(np.array(pose, np.float32).copy()); tracked.append(bool(ok))
    finally:
        O.set_perturbation(0)
    return (np.stack(poses), np.array(tracked), vol) if keep_volume else (np.stack(poses), np.array(tracked))


def divergence(pa, pb):
    """(worst |dt| over the translation components, worst |dR| over the rotation entries) between two pose sequences."""
    pa, pb = np.asarray(pa, np.float64), np.asarray(pb, np.float64)
    return float(np.max(np.abs(pa[:, :3, 3] - pb[:, :3, 3]))), float(np.max(np.abs(pa[:, :3, :3] - pb[:, :3, :3])))


def measure_floor(res, size, cam, n_frames, which=("reversed_sums", "exp2_taps", "fused_accumulation", "reciprocal_solve", "all"), **kw):
    base, tracked = oracle_sequence(res, size, cam, n_frames, 0, **kw)
    out = {"frames": n_frames, "all_tracked": bool(tracked.all()), "perturbations": {}}
    for name in which:
        p, t = oracle_sequence(res, size, cam, n_frames, PERTURBATIONS[name], **kw)
        dt, dr = divergence(base, p)
        out["perturbations"][name] = {"dt_m": dt, "dr": dr, "all_tracked": bool(t.all())}
    out["floor_dt_m"] = max(v["dt_m"] for v in out["perturbations"].values())
    out["floor_dr"] = max(v["dr"] for v in out["perturbations"].values())
    return base, out
