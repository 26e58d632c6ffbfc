"""Deterministic synthetic depth streams (SURVEY.md §8d "Scene S") and the stock parameter set.

The reference reads frames from a sensor or a TUM directory (src/DataSourceProducerRGBDDataset.cpp) and
delivers 16-bit millimetre depth, 0 = invalid (src/FrameData.h:57,70).  No dataset ships with it, so the
bench and the parity tests render an analytic scene in float64 and quantise it to that sensor contract.

Scene S: the interior of an axis-aligned box [m, size-m]^3 (m = 0.25*size) seen through its open front
face, plus a sphere of radius 0.15*size at the volume centre and four satellite spheres (radius 0.05*size) placed
asymmetrically in front of it.  The satellites are an addition to SURVEY.md's sketch: box + one sphere leaves one
translation almost unconstrained inside the 2 m integration gate, and the reference solves its 6x6 normal equations in
fp32 in world coordinates, so on that scene last-bit differences decide the pose (see DESIGN.md "Tracking parity").  Camera k = pose0 o small circular motion
(radius 2 cm, period 100 frames, yaw +-1 degree), pose0 = reference initial pose (src/HybKinectfu.cpp:51-54).
"""
import math

import numpy as np

# src/config.ini defaults, except the camera (VGA 525/319.5/239.5 per BASELINE.md) and what a config overrides
STOCK = dict(
    depth_trunc_min=0.3, depth_trunc_max=4.0, filter_sigma_pixel=2.0, filter_sigma_depth=0.03,
    volume_max_weight=128.0, integrate_sdf_trunc=0.05, integrate_depth_trunc=2.0, raycast_increment_factor=0.7,
    icp_thre_dist=0.1, icp_thre_sin_angle=0.1, camera_shake_dist=0.3, camera_shake_angle=0.3,
    sdf_max_iter_nums=6, max_triangle_num=6500000,
)


# (centre / size, radius / size): the central sphere of SURVEY.md's Scene S and four satellites
SPHERES = [(0.5, 0.5, 0.5, 0.15), (0.36, 0.40, 0.30, 0.05), (0.66, 0.37, 0.34, 0.05), (0.40, 0.65, 0.38, 0.05), (0.63, 0.62, 0.28, 0.05)]


def vga_camera(scale=1):
    """(cols, rows, cx, cy, fx, fy); scale=2 gives the 1280x960 camera of config C5."""
    return (640 * scale, 480 * scale, 319.5 * scale, 239.5 * scale, 525.0 * scale, 525.0 * scale)


def pose0(size, trunc_min=STOCK["depth_trunc_min"]):
    p = np.eye(4, dtype=np.float32)
    p[0, 3] = np.float32(size / 2.0)
    p[1, 3] = np.float32(size / 2.0)
    p[2, 3] = np.float32(-trunc_min)
    return p


def trajectory_pose(k, size, trunc_min=STOCK["depth_trunc_min"], radius=0.02, period=100, yaw_deg=1.0):
    """Ground-truth camera->world pose of frame k (float64)."""
    phi = 2.0 * math.pi * k / period
    yaw = math.radians(yaw_deg) * math.sin(phi)
    c, s = math.cos(yaw), math.sin(yaw)
    local = np.array([[c, 0, s, radius * (math.cos(phi) - 1.0)],
                      [0, 1, 0, radius * math.sin(phi)],
                      [-s, 0, c, 0.0],
                      [0, 0, 0, 1.0]])
    return pose0(size, trunc_min).astype(np.float64) @ local


def render_depth_mm(pose, cam, size, sphere=True, plane_depth=None):
    """u16 millimetre z-depth image of Scene S from camera->world `pose` (4x4)."""
    cols, rows, cx, cy, fx, fy = cam
    if plane_depth is not None:
        return np.full((rows, cols), int(round(plane_depth * 1000.0)), dtype=np.uint16)
    pose = np.asarray(pose, dtype=np.float64)
    u = (np.arange(cols, dtype=np.float64) - cx) / fx
    v = (np.arange(rows, dtype=np.float64) - cy) / fy
    uu, vv = np.meshgrid(u, v)
    dcam = np.stack([uu, vv, np.ones_like(uu)], axis=-1)            # camera-frame ray with z = 1 -> parameter = z-depth
    d = dcam @ pose[:3, :3].T
    o = pose[:3, 3]
    lo, hi = 0.25 * size, 0.75 * size
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (lo - o) / d
        t2 = (hi - o) / d
    tnear = np.nanmax(np.minimum(t1, t2), axis=-1)
    tfar = np.nanmin(np.maximum(t1, t2), axis=-1)
    hit = (tfar > np.maximum(tnear, 0.0))
    depth = np.where(hit, tfar, 0.0)                                  # interior wall = exit point of the box
    if sphere:
        a = np.sum(d * d, axis=-1)
        for (cx_, cy_, cz_, rr) in SPHERES:
            ctr = np.array([cx_, cy_, cz_]) * size
            r = rr * size
            oc = o - ctr
            b = 2.0 * (d @ oc)
            c = float(oc @ oc) - r * r
            disc = b * b - 4 * a * c
            with np.errstate(invalid="ignore"):
                ts = (-b - np.sqrt(disc)) / (2 * a)
            sph = (disc > 0) & (ts > 0)
            depth = np.where(sph & ((ts < depth) | (depth == 0)), ts, depth)
    mm = np.floor(depth * 1000.0 + 0.5)
    return np.clip(mm, 0, 65535).astype(np.uint16)


def make_stream(n_frames, cam, size, trunc_min=STOCK["depth_trunc_min"], start=0):
    """[n, rows, cols] u16 frames and their ground-truth poses."""
    frames = np.empty((n_frames, cam[1], cam[0]), dtype=np.uint16)
    poses = np.empty((n_frames, 4, 4), dtype=np.float64)
    for i in range(n_frames):
        poses[i] = trajectory_pose(start + i, size, trunc_min)
        frames[i] = render_depth_mm(poses[i], cam, size)
    return frames, poses


# ---- sensor noise ("no RNG needed unless noise is enabled -- then LCG seed 12345") -----------------------------
LCG_A, LCG_C, LCG_SEED = 1664525, 1013904223, 12345


def _lcg(state):
    return (state * np.uint64(LCG_A) + np.uint64(LCG_C)) & np.uint64(0xFFFFFFFF)


def add_sensor_noise(frames_mm, hole_fraction=0.02, seed=LCG_SEED, first_frame=0):
    """Depth-sensor noise on u16-millimetre frames, deterministic: every pixel of every frame draws from the 32-bit LCG (1664525, 1013904223)
    started at seed + its global index.  Axial noise with the Kinect's depth dependence, sigma(z) = 1.2 mm + 1.9 mm/m^2 (z - 0.4 m)^2 (sum of
    four uniform draws ~ normal), re-quantised to whole millimetres; `hole_fraction` of the valid pixels drop out (0 = invalid,
    src/FrameData.h:57).  The bilateral filter's early return (DataPreprocesser.cu:66-69), the ICP's rejection tests and the fusion pass's
    partial waves all see different inputs than on the noise-free scene."""
    f = np.ascontiguousarray(frames_mm, np.uint16)
    shape = f.shape
    n = f.size
    idx = np.arange(n, dtype=np.uint64) + np.uint64(first_frame) * np.uint64(shape[-1] * shape[-2])
    st = _lcg((np.uint64(seed) + idx * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF))
    acc = np.zeros(n, np.float64)
    for _ in range(4):
        st = _lcg(st)
        acc += (st >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)
    g = (acc - 2.0) * math.sqrt(3.0)                                  # four uniforms: variance 4/12 -> unit variance
    st = _lcg(st)
    drop = (st >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0) < hole_fraction
    z = f.reshape(-1).astype(np.float64) * 1e-3
    sigma = 0.0012 + 0.0019 * (z - 0.4) ** 2
    out = np.floor((z + sigma * g) * 1000.0 + 0.5)
    out = np.where((f.reshape(-1) == 0) | drop, 0.0, np.clip(out, 1, 65535))
    return out.astype(np.uint16).reshape(shape)
