"""Pose comparison helpers (numpy only) shared by bench.py's parity witness and the tests.

The tracked pose is a chain of fp32 compositions `cur = T * cur` that is never re-orthonormalised (here as in the reference,
src/CameraPoseFinderICP.cpp:143), so after a few thousand products its rotation block sits ~1e-6 off SO(3).  Measuring the angle
between two such blocks with arccos((trace(Ra^T Rb) - 1) / 2) turns that defect into sqrt(2 * defect) ~ 2e-3 "rad" (or clips to 0):
the expression is ill-conditioned at 0 and measures |R|_F^2 - 3 there, not an angle.  `rotation_angle` projects both blocks onto SO(3)
first and reads the angle from the skew part (atan2 form: well conditioned at 0 and at pi).
"""
import numpy as np


def project_to_so3(r):
    """nearest rotation matrix (Frobenius) to the 3x3 block r: U V^T of its SVD, determinant forced to +1"""
    u, _, vt = np.linalg.svd(np.asarray(r, np.float64).reshape(3, 3))
    d = np.sign(np.linalg.det(u @ vt))
    return u @ np.diag([1.0, 1.0, d if d != 0 else 1.0]) @ vt


def orthonormality_defect(r):
    """|R^T R - I|_F of a 3x3 block: how far a composed fp32 pose has drifted off SO(3)"""
    r = np.asarray(r, np.float64).reshape(3, 3)
    return float(np.linalg.norm(r.T @ r - np.eye(3)))


def rotation_angle(ra, rb):
    """angle (rad) of the relative rotation between two 3x3 blocks, each projected to SO(3) first:
    theta = atan2(|vee(D - D^T)| / 2, (tr D - 1) / 2), D = Ra^T Rb"""
    d = project_to_so3(ra).T @ project_to_so3(rb)
    s = 0.5 * np.linalg.norm([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]])
    c = 0.5 * (np.trace(d) - 1.0)
    return float(np.arctan2(s, c))


def pose_difference(pa, pb):
    """(translation distance, rotation angle, max |element difference| of the rotation blocks) of two 4x4 poses"""
    a, b = np.asarray(pa, np.float64).reshape(4, 4), np.asarray(pb, np.float64).reshape(4, 4)
    return (float(np.linalg.norm(a[:3, 3] - b[:3, 3])), rotation_angle(a[:3, :3], b[:3, :3]),
            float(np.max(np.abs(a[:3, :3] - b[:3, :3]))))
