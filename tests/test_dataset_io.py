"""CPU: the callers / data formats either side of the per-frame path -- PNG decode, TUM RGB-D directory
reader, nearest-in-time association, trajectory writer, quaternion <-> pose -- through the host library's GPU-free entry points.
Each check has an independent Python restatement of what the reference computes (file:line cited), not a re-run of our code."""
import os

import numpy as np
import pytest
from scipy import ndimage
from scipy.spatial.transform import Rotation

from hybkinectfu_amd import host_app as H
from hybkinectfu_amd import tum


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(7)


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4, None])
def test_png_roundtrip_all_filters(tmp_path, rng, filter_type):
    d16 = rng.integers(0, 65536, size=(37, 53), dtype=np.uint16)
    rgb = rng.integers(0, 256, size=(29, 41, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, size=(8, 9, 4), dtype=np.uint8)
    g8 = rng.integers(0, 256, size=(5, 7), dtype=np.uint8)
    for name, img in (("d16", d16), ("rgb", rgb), ("rgba", rgba), ("g8", g8)):
        path = str(tmp_path / (name + ".png"))
        tum.write_png(path, img, filter_type=filter_type, idat_split=3 if name == "rgb" else 0)
        got = H.png_read(path)
        assert got is not None
        assert np.array_equal(got.reshape(img.shape), img), name


def test_png_written_by_another_encoder(tmp_path, rng):
    PIL = pytest.importorskip("PIL.Image")
    d16 = rng.integers(0, 65536, size=(48, 64), dtype=np.uint16)
    rgb = rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
    PIL.fromarray(d16).save(str(tmp_path / "d.png"))            # mode I;16, adaptive filters, its own zlib settings
    PIL.fromarray(rgb).save(str(tmp_path / "c.png"), optimize=True)
    assert np.array_equal(H.png_read(str(tmp_path / "d.png"))[..., 0], d16)
    assert np.array_equal(H.png_read(str(tmp_path / "c.png")), rgb)


def test_png_rejects_damage(tmp_path, rng):
    path = str(tmp_path / "x.png")
    tum.write_png(path, rng.integers(0, 65536, size=(16, 16), dtype=np.uint16))
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0x55                                # CRC of the IDAT chunk no longer matches
    open(path, "wb").write(bytes(blob))
    assert H.png_read(path) is None
    assert H.png_read(str(tmp_path / "missing.png")) is None


def ref_nearest_sequence(stamps, targets):
    """DataSourceProducerRGBDDataset::parseFrameLineColor / CameraPoseFinderFromFile::parseFrameFromFile (.cpp:67-99 / :34-65):
    a file cursor that only moves forward; the first row at or after the target competes with the row read just before it in
    the same call (time 0 when there is none), and is pushed back when it loses."""
    cur, out = 0, []
    for tgt in targets:
        last, res = 0.0, -1.0
        while cur < len(stamps):
            at = cur
            s = stamps[cur]; cur += 1
            if s >= tgt:
                if s - tgt > tgt - last:
                    res, cur = last, at
                else:
                    res = s
                break
            last = s
        out.append(res)
    return np.array(out)


def test_nearest_association_matches_reference_cursor(tmp_path, rng):
    for trial in range(20):
        stamps = np.cumsum(rng.uniform(0.01, 0.05, size=60)) + 100.0
        targets = np.sort(rng.uniform(99.9, stamps[-1] + 0.2, size=40))
        path = str(tmp_path / ("list%d.txt" % trial))
        with open(path, "w") as f:
            f.write("# a\n# b\n# c\n")
            for s in stamps:
                f.write("%.6f rgb/%.6f.png\n" % (s, s))
        listed = np.array([float("%.6f" % s) for s in stamps])
        got = H.table_nearest(path, 3, targets)
        assert np.array_equal(got, ref_nearest_sequence(listed, targets))


def cv_pyrdown_u16(img):
    """cv::pyrDown on CV_16U: separable [1 4 6 4 1] with BORDER_REFLECT_101, integer sums, (s + 128) >> 8, every other pixel."""
    k = np.array([1, 4, 6, 4, 1], np.int64)
    a = ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")
    a = ndimage.correlate1d(a, k, axis=0, mode="mirror")
    return ((a[::2, ::2] + 128) >> 8).astype(np.uint16)


def test_pyrdown16_matches_opencv_integer_path(rng):
    for shape in ((96, 128), (31, 45), (2, 2), (1, 9)):
        img = rng.integers(0, 65536, size=shape, dtype=np.uint16)
        assert np.array_equal(H.pyrdown16(img), cv_pyrdown_u16(img)), shape


def test_tum_directory_reader(tmp_path, rng):
    cols, rows, n = 64, 48, 6
    mm = rng.integers(0, 9000, size=(n, rows, cols)).astype(np.uint16)
    mm[:, :4] = 0
    rgb = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    d = str(tmp_path / "seq") + "/"
    stamps = tum.write_dataset(d, mm, rgb_frames=rgb, raw_scale=5)
    depth, _, ds, _ = H.dataset_read(d, cols, rows, n + 3)
    assert depth.shape[0] == n                                  # stops at the end of depth.txt
    assert np.array_equal(depth, mm)                            # raw / 5 (DataSourceProducerRGBDDataset.cpp:103)
    assert np.allclose(ds, stamps, atol=1e-6)
    depth, bgr, ds, cs = H.dataset_read(d, cols, rows, n + 3, with_color=True)
    # every colour image is 4 ms OLDER than its depth frame, so the last depth frame has no colour row at or after it and the
    # reference's reader gives up there (parseFrameLineColor returns false at end of file, :98)
    assert depth.shape[0] == n - 1
    assert np.array_equal(depth, mm[:n - 1])
    assert np.array_equal(bgr, rgb[:n - 1, ..., ::-1])          # cv::imread order; the nearer (earlier) image wins, the later is pushed back
    assert np.allclose(ds, stamps[:n - 1], atol=1e-6) and np.allclose(cs, np.array(stamps[:n - 1]) - 0.004, atol=1e-6)
    # raw values that are not multiples of five round to the nearest millimetre
    raw = rng.integers(0, 65536, size=(1, rows, cols)).astype(np.uint16)
    d2 = str(tmp_path / "seq2") + "/"
    tum.write_dataset(d2, raw, raw_scale=1)
    got, _, _, _ = H.dataset_read(d2, cols, rows, 1)
    assert np.array_equal(got[0], np.rint(raw[0].astype(np.float64) / 5.0).astype(np.uint16))
    # a sensor image twice the configured size is halved once (:105-113)
    big = rng.integers(0, 9000, size=(1, 2 * rows, 2 * cols)).astype(np.uint16)
    d3 = str(tmp_path / "seq3") + "/"
    tum.write_dataset(d3, big, raw_scale=5)
    got, _, _, _ = H.dataset_read(d3, cols, rows, 1)
    assert np.array_equal(got[0], cv_pyrdown_u16(big[0]))
    with pytest.raises(Exception):
        H.dataset_read(str(tmp_path / "nowhere") + "/", cols, rows, 1)


def test_quaternion_conversions(rng):
    rots = list(Rotation.random(200, random_state=3).as_matrix())
    rots += [np.eye(3), Rotation.from_euler("x", 179.9, degrees=True).as_matrix(), Rotation.from_euler("y", 180, degrees=True).as_matrix(),
             Rotation.from_euler("z", -179.5, degrees=True).as_matrix()]      # trace <= 0: the three largest-diagonal branches
    for R in rots:
        pose = np.eye(4, dtype=np.float32); pose[:3, :3] = R; pose[:3, 3] = rng.uniform(-2, 2, 3)
        q = H.quat_from_pose(pose)
        ref = Rotation.from_matrix(R).as_quat()
        if np.dot(ref, q) < 0:
            ref = -ref
        assert np.allclose(q, ref, atol=2e-6)
        back = H.pose_from_quat(pose[:3, 3], q)
        assert np.allclose(back, pose, atol=2e-6)


def test_trajectory_writer_format(tmp_path, rng):
    n = 5
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    for k in range(n):
        poses[k, :3, :3] = Rotation.from_rotvec(rng.uniform(-1, 1, 3)).as_matrix()
        poses[k, :3, 3] = rng.uniform(-3, 3, 3)
    stamps = 1305031102.175304 + np.arange(n) / 30.0
    path = str(tmp_path / "traj.txt")
    assert H.trajectory_write(path, poses, stamps) == n
    lines = open(path).read().splitlines()
    assert lines[0] == "# trajectory" and lines[1] == "# file: " + path and lines[2] == "# timestamp tx ty tz qx qy qz qw"   # TrajectoryRecorder.cpp:12-14
    for k in range(n):
        q = H.quat_from_pose(poses[k])
        # `<< setprecision(14) << double`, `<< setprecision(6) << float` (:37-39): default floatfield == printf %g
        want = "%.14g %.6g %.6g %.6g %.6g %.6g %.6g %.6g" % ((stamps[k],) + tuple(float(v) for v in poses[k, :3, 3]) + tuple(float(v) for v in q))
        assert lines[3 + k] == want
