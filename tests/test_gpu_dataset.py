"""GPU: the dataset reader, the trajectory recorder and the file-driven pose finder around the per-frame path
(src/MainController.cpp:33-48 with useDatasetRGBD / recordTrajectory / useTrajFromFile switched on)."""
import numpy as np
import pytest

from hybkinectfu_amd import host_app as H
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import scene as S
from hybkinectfu_amd import tum

pytestmark = pytest.mark.gpu
P = S.STOCK


def make_sequence(tmp_path, n, cam, size):
    frames = [S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)]
    poses = [S.trajectory_pose(k, size) for k in range(n)]
    d = str(tmp_path / "seq") + "/"
    stamps = tum.write_dataset(d, frames, poses=poses)
    return d, frames, poses, stamps


def test_dataset_frames_tracked_and_recorded(tmp_path):
    res, size, cam, n = 128, 3.0, S.vga_camera(), 4
    trunc = 5 * size / res
    d, frames, _, stamps = make_sequence(tmp_path, n, cam, size)
    out = str(tmp_path / "trajectory.txt")
    # the same frames handed over in memory: the directory reader must change nothing
    app = H.App(res, size, cam, sdf_trunc=trunc)
    want = []
    for k in range(n):
        assert app.process_frame(frames[k], k)
        want.append(app.pose()[1].copy())
    app.close()
    app = H.App(res, size, cam, sdf_trunc=trunc, dataset_dir=d, traj_write=out)
    got = []
    for k in range(n):
        r = app.process_dataset_frame(k)
        assert r is not None and r[0] and abs(r[1] - stamps[k]) < 1e-6
        got.append(app.pose()[1].copy())
    assert app.process_dataset_frame(n) is None                  # end of depth.txt
    app.close()
    for k in range(n):
        assert np.array_equal(got[k], want[k]), k
    lines = open(out).read().splitlines()
    assert len(lines) == 3 + n                                   # header + one pose per tracked frame (HybKinectfu.cpp:129-132)
    for k in range(n):
        q = H.quat_from_pose(got[k])
        assert lines[3 + k] == "%.14g %.6g %.6g %.6g %.6g %.6g %.6g %.6g" % (
            (float("%.6f" % stamps[k]),) + tuple(float(v) for v in got[k][:3, 3]) + tuple(float(v) for v in q))
        # ... and independently of the library's own converter: the written quaternion (x y z w, TrajectoryRecorder.cpp) is the pose's rotation
        from scipy.spatial.transform import Rotation
        fq = np.array([float(v) for v in lines[3 + k].split()[4:8]])
        ref = Rotation.from_matrix(got[k][:3, :3].astype(np.float64)).as_quat()
        assert min(np.max(np.abs(fq - ref)), np.max(np.abs(fq + ref))) < 5e-6, (k, fq, ref)


def test_pose_finder_from_file_drives_the_fusion(tmp_path):
    res, size, cam, n = 128, 3.0, S.vga_camera(), 4
    trunc = 5 * size / res
    d, frames, gt, stamps = make_sequence(tmp_path, n, cam, size)
    app = H.App(res, size, cam, sdf_trunc=trunc, dataset_dir=d, traj_read=d + "groundtruth.txt")
    poses = []
    for k in range(n):
        r = app.process_dataset_frame(k)
        assert r is not None and r[0]
        poses.append(app.pose()[1].copy())
    kcam = K.camera(*cam)
    vol_app = K.Context.borrow(app.ctx_handle(), kcam, res, size).download_volume()
    app.close()
    # CameraPoseFinderFromFile.cpp:82-87: frame 0 keeps the initial pose, later frames are re-based on it
    file_pose = [H.pose_from_quat(*_file_row(d, k)) for k in range(n)]
    from scipy.spatial.transform import Rotation
    for k in range(n):                                           # the library's quaternion -> pose against scipy's
        t, q = _file_row(d, k)
        assert np.allclose(file_pose[k][:3, :3], Rotation.from_quat(np.asarray(q, np.float64)).as_matrix(), atol=1e-5) and np.allclose(file_pose[k][:3, 3], t, atol=1e-7)
    refer = poses[0].astype(np.float64) @ np.linalg.inv(file_pose[0].astype(np.float64))
    assert np.array_equal(poses[0], S.pose0(size).astype(np.float32))
    for k in range(1, n):
        assert np.allclose(poses[k], refer @ file_pose[k], atol=2e-6), k
        assert np.allclose(poses[k], S.pose0(size) @ np.linalg.inv(gt[0]) @ gt[k], atol=2e-5), k   # i.e. the scene's own motion
    # the fusion the application ran with those poses == the C ABI called directly with the same poses
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    for k in range(n):
        ctx.upload_depth_mm(frames[k])
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ctx.integrate(poses[k], trunc, P["integrate_depth_trunc"])
    vol = ctx.download_volume()
    ctx.close()
    assert np.array_equal(vol_app[1], vol[1]) and np.array_equal(vol_app[0], vol[0])
    assert (vol[1] > 0).sum() > 10000


def _file_row(d, k):
    rows = [l.split() for l in open(d + "groundtruth.txt").read().splitlines()[3:]]
    v = np.array(rows[k][1:], np.float32)
    return v[:3], v[3:7]
