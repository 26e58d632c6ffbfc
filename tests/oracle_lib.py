"""ctypes binding of the CPU oracle (oracle/libkforacle.so) and of oracle/_ref/libkfref.so.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("KF_ORACLE_SO") or os.path.join(ORACLE_DIR, "libkforacle.so")     # KF_ORACLE_SO: the sanitizer build
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libkfref.so")


class Cam(C.Structure):
    _fields_ = [("cols", C.c_uint32), ("rows", C.c_uint32), ("cx", C.c_float), ("cy", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float)]

    @staticmethod
    def make(cols, rows, cx, cy, fx, fy):
        return Cam(cols, rows, cx, cy, fx, fy)

    def half(self):
        """src/CameraPoseFinderICP.cpp:39-45"""
        f = np.float32
        return Cam(self.cols // 2, self.rows // 2, f(self.cx) / f(2), f(self.cy) / f(2), f(self.fx) / f(2), f(self.fy) / f(2))


class Volume(C.Structure):
    _fields_ = [("data", C.c_void_p), ("res", C.c_int32), ("size", C.c_float), ("max_weight", C.c_float),
                ("z_base", C.c_int32), ("z_layers", C.c_int32)]


VOXEL_DTYPE = np.dtype([("tsdf", "<f4"), ("weight", "<f4"), ("color", "u1", (3,)), ("pad", "u1")])
VERTEX_DTYPE = np.dtype([("pos", "<f4", (3,)), ("color", "<f4", (3,))])
TRI_DTYPE = np.dtype([("v", VERTEX_DTYPE, (3,))])
assert VOXEL_DTYPE.itemsize == 12 and TRI_DTYPE.itemsize == 72


def build_oracle():
    if os.environ.get("KF_ORACLE_SO"):
        return ORACLE_SO
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "kf_oracle.cpp")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libkforacle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_lib = None


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        _lib.okf_integrate.restype = C.c_uint64
        _lib.okf_count_weight_gt0.restype = C.c_uint64
        _lib.okf_marching_cubes.restype = C.c_uint32
    return _lib


def m16(m):
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(16))


class OVolume:
    """Dense 12-byte-voxel volume for the oracle.  band=(z_base, z_layers): only those layers of the res^3 grid are held (volumes too
    large for host memory: 2048^3 is 103 GB); reads outside the band count as violations (band_violations())."""

    def __init__(self, res, size, max_weight, band=None):
        self.res, self.size, self.max_weight = int(res), float(size), float(max_weight)
        self.z_base, self.layers = (0, self.res) if band is None else (int(band[0]), int(band[1]))
        self.vox = np.zeros(self.res * self.res * self.layers, dtype=VOXEL_DTYPE)
        self.c = Volume(self.vox.ctypes.data, self.res, self.size, self.max_weight, self.z_base, 0 if band is None else self.layers)

    @property
    def tsdf(self):
        return self.vox["tsdf"].reshape(self.layers, self.res, self.res)   # [z - z_base][y][x]

    @property
    def weight(self):
        return self.vox["weight"].reshape(self.layers, self.res, self.res)

    @property
    def color(self):
        return self.vox["color"].reshape(self.layers, self.res, self.res, 3)


def band_violations():
    """reads outside a z-band volume's layers since the last call (must be 0 for a test that claims to stay inside its band)"""
    lib().okf_band_violations.restype = C.c_uint64
    return int(lib().okf_band_violations())


def depth_mm_to_m(mm):
    mm = np.ascontiguousarray(mm, dtype=np.uint16)
    out = np.empty(mm.shape, np.float32)
    lib().okf_depth_mm_to_m(fp(mm), C.c_int(mm.size), fp(out))
    return out


def trunc_depth(d, tmin, tmax):
    out = np.empty_like(d)
    lib().okf_trunc_depth(fp(d), C.c_int(d.shape[1]), C.c_int(d.shape[0]), C.c_float(tmin), C.c_float(tmax), fp(out))
    return out


def bilateral(d, sigma_pixel, sigma_depth):
    out = np.empty_like(d)
    lib().okf_bilateral(fp(d), C.c_int(d.shape[1]), C.c_int(d.shape[0]), C.c_float(sigma_pixel), C.c_float(sigma_depth), fp(out))
    return out


def depth_to_vertices(d, cam):
    out = np.empty(d.shape + (4,), np.float32)
    lib().okf_depth_to_vertices(fp(d), C.byref(cam), fp(out))
    return out


def vertices_to_normals(v):
    out = np.empty_like(v)
    lib().okf_vertices_to_normals(fp(v), C.c_int(v.shape[1]), C.c_int(v.shape[0]), fp(out))
    return out


def pyrdown(v, normals=False):
    out = np.empty((v.shape[0] // 2, v.shape[1] // 2, 4), np.float32)
    f = lib().okf_pyrdown_normals if normals else lib().okf_pyrdown_vertices
    f(fp(v), C.c_int(v.shape[1]), C.c_int(v.shape[0]), fp(out))
    return out


def pyramid(v0, levels, normals=False):
    out = [np.ascontiguousarray(v0)]
    for _ in range(1, levels):
        out.append(pyrdown(out[-1], normals))
    return out


def icp_system(new_v, new_n, model_v, model_n, cam, cur, last_inv, dist, sin):
    d = np.zeros(27, np.float64)
    f = np.zeros(27, np.float32)
    n = C.c_int(0)
    lib().okf_icp_system(fp(new_v), fp(new_n), fp(model_v), fp(model_n), C.byref(cam), fp(m16(cur)), fp(m16(last_inv)),
                         C.c_float(dist), C.c_float(sin), fp(d), fp(f), C.byref(n))
    return d, f, n.value


def icp_estimate(new_v, new_n, model_v, model_n, cam0, dist, sin, dist_shake, angle_shake, pose):
    levels = len(new_v)
    arr = lambda lst: (C.c_void_p * levels)(*[a.ctypes.data for a in lst])
    p = m16(pose).copy()
    ok = lib().okf_icp_estimate(arr(new_v), arr(new_n), arr(model_v), arr(model_n), C.c_int(levels), C.byref(cam0),
                                C.c_float(dist), C.c_float(sin), C.c_float(dist_shake), C.c_float(angle_shake), fp(p))
    return bool(ok), p.reshape(4, 4)


def solve6(sums27, check_det=True):
    x = np.zeros(6, np.float32)
    ok = lib().okf_solve6(fp(np.ascontiguousarray(sums27, np.float32)), C.c_int(int(check_det)), fp(x))
    return bool(ok), x


def vector6_to_transform(x, dist_shake, angle_shake):
    t = np.zeros(16, np.float32)
    ok = lib().okf_vector6_to_transform(fp(np.ascontiguousarray(x, np.float32)), C.c_float(dist_shake), C.c_float(angle_shake), fp(t))
    return bool(ok), t.reshape(4, 4)


def mat44_inverse(m):
    out = np.zeros(16, np.float32)
    lib().okf_mat44_inverse(fp(m16(m)), fp(out))
    return out.reshape(4, 4)


def mat44_mul(a, b):
    out = np.zeros(16, np.float32)
    lib().okf_mat44_mul(fp(m16(a)), fp(m16(b)), fp(out))
    return out.reshape(4, 4)


def exp_map(v6):
    r = np.zeros(9, np.float64)
    t = np.zeros(3, np.float64)
    lib().okf_exp_map(fp(np.ascontiguousarray(v6, np.float64)), fp(r), fp(t))
    return r.reshape(3, 3), t


def sdf_system(vol, depth, cam, cur):
    d = np.zeros(27, np.float64)
    f = np.zeros(27, np.float32)
    n = C.c_int(0)
    lib().okf_sdf_system(C.byref(vol.c), fp(depth), C.byref(cam), fp(m16(cur)), fp(d), fp(f), C.byref(n))
    return d, f, n.value


def sdf_estimate(vol, depth, cam, max_iter, dist_shake, angle_shake, pose):
    p = m16(pose).copy()
    it = C.c_int(0)
    ok = lib().okf_sdf_estimate(C.byref(vol.c), fp(depth), C.byref(cam), C.c_int(max_iter), C.c_float(dist_shake),
                                C.c_float(angle_shake), fp(p), C.byref(it))
    return bool(ok), p.reshape(4, 4), it.value


def integrate(vol, depth, normals, rgb, has_color, color_angled, pose, sdf_trunc, max_dist, depth_cam, rgb_cam, z0=0, z1=None):
    z1 = vol.res if z1 is None else z1
    rgbp = fp(rgb) if rgb is not None else None
    return int(lib().okf_integrate(C.byref(vol.c), C.c_int(z0), C.c_int(z1), fp(depth), fp(normals), rgbp, C.c_int(int(has_color)),
                                   C.c_int(int(color_angled)), fp(m16(pose)), C.c_float(sdf_trunc), C.c_float(max_dist),
                                   C.byref(depth_cam), C.byref(rgb_cam)))


def count_weight_gt0(vol):
    return int(lib().okf_count_weight_gt0(C.byref(vol.c)))


def raycast(vol, has_color, pose, inc, cam, near, far, want_steps=False):
    v = np.empty((cam.rows, cam.cols, 4), np.float32)
    n = np.empty((cam.rows, cam.cols, 4), np.float32)
    rgb = np.zeros((cam.rows, cam.cols, 3), np.uint8)
    steps = np.zeros((cam.rows, cam.cols), np.uint32) if want_steps else None
    lib().okf_raycast(C.byref(vol.c), C.c_int(int(has_color)), fp(m16(pose)), C.c_float(inc), C.byref(cam), C.c_float(near),
                      C.c_float(far), fp(v), fp(n), fp(rgb), fp(steps) if want_steps else None)
    return (v, n, rgb, steps) if want_steps else (v, n, rgb)


def marching_cubes(vol, has_color, thr, max_tris, z0=0, z1=None):
    z1 = vol.res if z1 is None else z1
    tris = np.zeros(max_tris, dtype=TRI_DTYPE)
    n = lib().okf_marching_cubes(C.byref(vol.c), C.c_int(z0), C.c_int(z1), C.c_int(int(has_color)), C.c_float(thr), fp(tris), C.c_uint32(max_tris))
    return tris[:n]


def interpolate_sdf(vol, pos):
    d = C.c_float(0)
    ok = lib().okf_interpolate_sdf(C.byref(vol.c), fp(np.ascontiguousarray(pos, np.float32)), C.byref(d))
    return bool(ok), d.value


def interpolate_color(vol, pos):
    out = np.zeros(3, np.uint8)
    ok = lib().okf_interpolate_color(C.byref(vol.c), fp(np.ascontiguousarray(pos, np.float32)), fp(out))
    return bool(ok), out


def world_to_voxel(vol, pos):
    out = np.zeros(3, np.int32)
    lib().okf_world_to_voxel(C.byref(vol.c), fp(np.ascontiguousarray(pos, np.float32)), fp(out))
    return out


def norm(v):
    lib().okf_norm.restype = C.c_float
    return lib().okf_norm(fp(np.ascontiguousarray(v, np.float32)))


def to_int(v):
    return lib().okf_to_int(C.c_double(v))


def set_threads(n):
    return lib().okf_set_threads(C.c_int(n))


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        # lazy binding: the never-executed GPU branches of DataMap.h reference cudaMalloc/cudaFree, which no library here provides
        _ref = C.CDLL(REF_SO, mode=os.RTLD_LAZY)
        _ref.ref_vol_create.restype = C.c_void_p
        _ref.ref_norm.restype = C.c_float
    return _ref


def set_perturbation(mode):
    """Tests only (tests/test_tracking_floor.py): last-bit perturbations of the oracle's tracker arithmetic -- bit 0 reversed
    summation order of the 27 ICP sums, bit 1 exp2f bilateral taps, bit 2 fused accumulation.  0 restores the restatement proper."""
    lib().okf_set_perturbation(C.c_int(int(mode)))
