"""hybkinectfu_amd/posemath.py: the rotation metric of bench.py's parity witness.
The tracked pose is a product of thousands of fp32 matrices and sits ~1e-6 off SO(3); arccos of the trace read that defect as ~2e-3 rad."""
import numpy as np

from hybkinectfu_amd import posemath as PM


def _rot(axis, ang):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    k = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * k + (1 - np.cos(ang)) * (k @ k)


def _drifted_fp32(n=2850, seed=3):
    """a rotation block after n fp32 compositions with small increments (149 frames x 19 Gauss-Newton steps), never re-orthonormalised"""
    rng = np.random.default_rng(seed)
    r = _rot([0.2, 1.0, -0.3], 0.4).astype(np.float32)
    for _ in range(n):
        inc = _rot(rng.normal(size=3), 2e-4 * rng.random()).astype(np.float32)
        r = (inc @ r).astype(np.float32)
    return r


def test_equal_non_orthonormal_fp32_blocks_give_zero():
    r = _drifted_fp32()
    assert PM.orthonormality_defect(r) > 1e-7          # it HAS drifted: the case the old metric misread
    assert PM.rotation_angle(r, r.copy()) == 0.0
    # ... and the old expression on the same input is not an angle: it reads the defect (or clips to exactly 0)
    old = np.arccos(np.clip((np.trace(r.astype(np.float64).T @ r.astype(np.float64)) - 1) / 2, -1, 1))
    assert old == 0.0 or old > 1e-4


def test_known_small_offset_is_recovered_on_drifted_blocks():
    r = _drifted_fp32().astype(np.float64)
    for ang in (5e-4, 1e-4, 3e-6):
        rb = _rot([0.3, -0.5, 0.8], ang) @ r
        got = PM.rotation_angle(r, rb)
        assert abs(got - ang) < 2e-7 + 1e-3 * ang, (ang, got)
        # after an fp32 round trip of both (what kf_get_pose hands back) the reading moves by fp32 round-off only
        got32 = PM.rotation_angle(r.astype(np.float32), rb.astype(np.float32))
        assert abs(got32 - ang) < 5e-7, (ang, got32)


def test_large_angles_and_pi():
    a = _rot([1, 2, 3], 0.3)
    for ang in (0.5, 2.0, np.pi - 1e-6, np.pi):
        assert abs(PM.rotation_angle(a, _rot([0, 0, 1], ang) @ a) - ang) < 1e-9


def test_pose_difference_parts():
    pa = np.eye(4); pb = np.eye(4)
    pb[:3, :3] = _rot([0, 1, 0], 2e-4); pb[:3, 3] = [3e-5, 0, 4e-5]
    d_t, ang, d_el = PM.pose_difference(pa, pb)
    assert abs(d_t - 5e-5) < 1e-12 and abs(ang - 2e-4) < 1e-10 and abs(d_el - np.sin(2e-4)) < 1e-10


def test_projection_is_a_rotation_and_idempotent():
    r = _drifted_fp32()
    q = PM.project_to_so3(r)
    assert PM.orthonormality_defect(q) < 1e-14 and abs(np.linalg.det(q) - 1) < 1e-14
    assert np.max(np.abs(PM.project_to_so3(q) - q)) < 1e-15
