"""ctypes binding of the C ABI in include/hybkf.h (libhybkf.so).

Python here is plumbing for tests and bench.py; the product is the shared library and the C++ host classes in
hybkinectfu_amd/host/.  There is NO CPU fallback: if the HIP library is missing, loading raises.
"""
import ctypes as C
import os
import weakref
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# KF_LIB: tools/ may point at the experiments variant (libhybkf_exp.so, `make -C csrc experiments`); default = the product library
LIB_PATH = os.environ.get("KF_LIB") or os.path.join(PKG_DIR, "libhybkf.so")


class CameraParams(C.Structure):
    _fields_ = [("cols", C.c_uint32), ("rows", C.c_uint32), ("cx", C.c_float), ("cy", C.c_float), ("fx", C.c_float), ("fy", C.c_float)]


class Mat44(C.Structure):
    _fields_ = [("m", C.c_float * 16)]

    @staticmethod
    def of(a):
        a = np.asarray(a, dtype=np.float32).reshape(16)
        return Mat44((C.c_float * 16)(*a.tolist()))

    def numpy(self):
        return np.array(list(self.m), dtype=np.float32).reshape(4, 4)


class IntegrateParams(C.Structure):
    _fields_ = [("sdf_truncation", C.c_float), ("max_integrate_dist", C.c_float)]


class RaycastParams(C.Structure):
    _fields_ = [("ray_increment", C.c_float)]


class VolumeParams(C.Structure):
    _fields_ = [("resolution", C.c_uint32), ("size_m", C.c_float), ("max_weight", C.c_float)]


class IcpParams(C.Structure):
    _fields_ = [("pyramid_levels", C.c_uint32), ("norm_sin_thres", C.c_float), ("dist_thres", C.c_float),
                ("dist_shake", C.c_float), ("angle_shake", C.c_float)]


class SdfTrackerParams(C.Structure):
    _fields_ = [("max_iter_nums", C.c_uint32), ("dist_shake", C.c_float), ("angle_shake", C.c_float)]


class Config(C.Structure):
    _fields_ = [("depth_camera", CameraParams), ("rgb_camera", CameraParams), ("volume", VolumeParams),
                ("pyramid_levels", C.c_uint32), ("max_triangles", C.c_uint32), ("has_color", C.c_int32), ("device", C.c_int32),
                ("slab_z_begin", C.c_uint32), ("slab_z_end", C.c_uint32), ("slab_halo", C.c_uint32)]


class TrackResult(C.Structure):
    _fields_ = [("pose", Mat44), ("tracked", C.c_int32), ("status", C.c_int32), ("iterations", C.c_int32), ("launch_form", C.c_int32)]


class VolumeStats(C.Structure):
    _fields_ = [("updated_last", C.c_uint64), ("weight_gt0", C.c_uint64), ("bricks_active", C.c_uint64), ("bricks_total", C.c_uint64),
                ("updated_total", C.c_uint64), ("frames_fused", C.c_uint64), ("frames_lost", C.c_uint64)]


VERTEX_DTYPE = np.dtype([("pos", "<f4", (3,)), ("color", "<f4", (3,))])
TRI_DTYPE = np.dtype([("v", VERTEX_DTYPE, (3,))])

MAP_RAW_DEPTH, MAP_TRUNCED_DEPTH, MAP_FILTERED_DEPTH = 0, 1, 2
MAP_NEW_VERTICES, MAP_NEW_NORMALS, MAP_MODEL_VERTICES, MAP_MODEL_NORMALS = 3, 4, 5, 6
MAP_RAW_RGB, MAP_RAYCAST_RGB = 7, 8

# every symbol include/hybkf.h declares (tests check that the library exports each of them)
SYMBOLS = [
    "kf_error_string", "kf_version", "kf_create", "kf_destroy", "kf_synchronize", "kf_stream", "kf_reset_volume",
    "kf_upload_depth_mm", "kf_set_depth_mm_device", "kf_upload_rgb", "kf_trunc_depth", "kf_bilateral_filter_depth",
    "kf_calculate_new_vertices", "kf_calculate_new_normals", "kf_preprocess", "kf_prefetch_frame", "kf_downsample_new_vertices",
    "kf_downsample_new_normals", "kf_downsample_model_vertices", "kf_downsample_model_normals",
    "kf_cal_point_to_plane_solver_params", "kf_cal_sdf_solver_params", "kf_read_solver_params", "kf_set_pose",
    "kf_icp_track", "kf_sdf_track", "kf_read_track_result", "kf_request_track_result", "kf_wait_track_result", "kf_integrate_volume", "kf_raycast_volume",
    "kf_marching_cubes", "kf_clear_triangles", "kf_triangle_count", "kf_read_triangles", "kf_download_map",
    "kf_upload_map", "kf_download_volume", "kf_upload_volume", "kf_get_volume_stats", "kf_stored_z_range",
    "kf_stage_timers", "kf_read_stage_ms", "kf_read_work_counters", "kf_set_stream", "kf_raycast_volume_slab", "kf_slab_mask_candidates", "kf_set_model_maps_device", "kf_raycast_volume_slab_cross", "kf_slab_ray_normals", "kf_raycast_volume_slab_cross_spec", "kf_slab_ray_normals_spec", "kf_set_model_maps_rays", "kf_selftest_div",
    "kf_icp_partition_begin", "kf_icp_partition_steps", "kf_icp_partition_step", "kf_icp_partition_finish",
    "kf_sdf_partition_begin", "kf_sdf_partition_step", "kf_sdf_partition_finish", "kf_set_defer", "kf_inject_track_stall",
    "kf_download_volume_device", "kf_upload_volume_device", "kf_resize_slab", "kf_count_layer_work", "kf_read_layer_work",
    "kf_upload_depth_mm_next", "kf_take_next_depth", "kf_cull_tail_counts", "kf_count_observed_voxels", "kf_get_fusion_counters",
]


def build(force=False):
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(PKG_DIR, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j6"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load():
    """dlopen libhybkf.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libhybkf.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.kf_error_string.restype = C.c_char_p
        _lib.kf_version.restype = C.c_char_p
        _lib.kf_stream.restype = C.c_void_p
    return _lib


class KfError(RuntimeError):
    pass


def _chk(st, what):
    if st != 0:
        raise KfError("%s failed: %d (%s)" % (what, st, load().kf_error_string(st).decode()))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def camera(cols, rows, cx, cy, fx, fy):
    return CameraParams(int(cols), int(rows), cx, cy, fx, fy)


def half_camera(c):
    """src/CameraPoseFinderICP.cpp:39-45 in fp32"""
    f = np.float32
    return CameraParams(c.cols // 2, c.rows // 2, f(c.cx) / f(2), f(c.cy) / f(2), f(c.fx) / f(2), f(c.fy) / f(2))


_LIVE = weakref.WeakSet()


def live_contexts():
    """Contexts created through this module and not yet closed (a second live kf_ctx switches the persistent loops off, so a
    leaked one changes launch forms: test harnesses close what a failed test left open)."""
    return [c for c in list(_LIVE) if c.h]


class Context:
    """One kf_ctx (one GPU / one z-slab)."""

    def __init__(self, depth_cam, volume_res, volume_size, max_weight=128.0, levels=3, max_triangles=0, has_color=False,
                 rgb_cam=None, device=0, slab=None, halo=0):
        self.lib = load()
        self.cam = depth_cam
        self.rgb_cam = rgb_cam if rgb_cam is not None else depth_cam
        self.res, self.size, self.levels = int(volume_res), float(volume_size), int(levels)
        z0, z1 = slab if slab is not None else (0, volume_res)
        cfg = Config(depth_cam, self.rgb_cam, VolumeParams(volume_res, volume_size, max_weight), levels, max_triangles,
                     int(has_color), device, z0, z1, halo)
        self.h = C.c_void_p()
        _chk(self.lib.kf_create(C.byref(cfg), C.byref(self.h)), "kf_create")
        _LIVE.add(self)
        z0s, z1s = C.c_uint32(), C.c_uint32()
        _chk(self.lib.kf_stored_z_range(self.h, C.byref(z0s), C.byref(z1s)), "kf_stored_z_range")
        self.stored = (z0s.value, z1s.value)
        self.owned = (z0, z1)

    @classmethod
    def borrow(cls, handle, depth_cam, volume_res, volume_size, levels=3):
        """Wrap a kf_ctx owned by someone else (the C++ host application): same methods, close() leaves it alone."""
        self = cls.__new__(cls)
        self.lib = load()
        self.cam = self.rgb_cam = depth_cam
        self.res, self.size, self.levels = int(volume_res), float(volume_size), int(levels)
        self.h = handle if isinstance(handle, C.c_void_p) else C.c_void_p(handle)
        self.borrowed = True
        z0s, z1s = C.c_uint32(), C.c_uint32()
        _chk(self.lib.kf_stored_z_range(self.h, C.byref(z0s), C.byref(z1s)), "kf_stored_z_range")
        self.stored = (z0s.value, z1s.value)
        self.owned = self.stored
        return self

    def close(self):
        if self.h:
            if not getattr(self, "borrowed", False):
                self.lib.kf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads ----
    def upload_depth_mm(self, mm):
        mm = np.ascontiguousarray(mm, np.uint16)
        _chk(self.lib.kf_upload_depth_mm(self.h, _p(mm), mm.shape[1], mm.shape[0]), "kf_upload_depth_mm")
        self.sync()

    def upload_depth_mm_next(self, mm):
        """stage the NEXT frame while the current one is processed; returns its device address (for prefetch_frame)"""
        mm = np.ascontiguousarray(mm, np.uint16)
        dev = C.c_void_p()
        _chk(self.lib.kf_upload_depth_mm_next(self.h, _p(mm), mm.shape[1], mm.shape[0], C.byref(dev)), "kf_upload_depth_mm_next")
        return dev.value

    def take_next_depth(self):
        _chk(self.lib.kf_take_next_depth(self.h), "kf_take_next_depth")

    def set_depth_mm_device(self, dev_ptr):
        _chk(self.lib.kf_set_depth_mm_device(self.h, C.c_void_p(dev_ptr), self.cam.cols, self.cam.rows), "kf_set_depth_mm_device")

    def upload_rgb(self, bgr):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        _chk(self.lib.kf_upload_rgb(self.h, _p(bgr), bgr.shape[1], bgr.shape[0]), "kf_upload_rgb")

    def upload_map(self, map_id, level, arr):
        arr = np.ascontiguousarray(arr)
        _chk(self.lib.kf_upload_map(self.h, map_id, level, _p(arr), C.c_size_t(arr.nbytes)), "kf_upload_map")

    def download_map(self, map_id, level=0):
        cols, rows = self.cam.cols >> level, self.cam.rows >> level
        if map_id in (MAP_RAW_DEPTH, MAP_TRUNCED_DEPTH, MAP_FILTERED_DEPTH):
            out = np.empty((self.cam.rows, self.cam.cols), np.float32)
        elif map_id == MAP_RAW_RGB:
            out = np.empty((self.rgb_cam.rows, self.rgb_cam.cols, 3), np.uint8)
        elif map_id == MAP_RAYCAST_RGB:
            out = np.empty((self.cam.rows, self.cam.cols, 3), np.uint8)
        else:
            out = np.empty((rows, cols, 4), np.float32)
        _chk(self.lib.kf_download_map(self.h, map_id, level, _p(out), C.c_size_t(out.nbytes)), "kf_download_map")
        return out

    # ---- per-stage wrappers (1:1 with src/cuda/CudaWrappers.h) ----
    def trunc_depth(self, tmin, tmax):
        _chk(self.lib.kf_trunc_depth(self.h, C.c_float(tmin), C.c_float(tmax)), "kf_trunc_depth")

    def bilateral(self, sigma_pixel, sigma_depth):
        _chk(self.lib.kf_bilateral_filter_depth(self.h, C.c_float(sigma_pixel), C.c_float(sigma_depth)), "kf_bilateral_filter_depth")

    def calculate_new_vertices(self):
        _chk(self.lib.kf_calculate_new_vertices(self.h, C.byref(self.cam)), "kf_calculate_new_vertices")

    def calculate_new_normals(self):
        _chk(self.lib.kf_calculate_new_normals(self.h), "kf_calculate_new_normals")

    def preprocess(self, tmin, tmax, sigma_pixel, sigma_depth):
        _chk(self.lib.kf_preprocess(self.h, C.c_float(tmin), C.c_float(tmax), C.c_float(sigma_pixel), C.c_float(sigma_depth),
                                    C.byref(self.cam)), "kf_preprocess")

    def prefetch_frame(self, dev_ptr, tmin, tmax, sigma_pixel, sigma_depth):
        _chk(self.lib.kf_prefetch_frame(self.h, C.c_void_p(dev_ptr), self.cam.cols, self.cam.rows, C.c_float(tmin), C.c_float(tmax),
                                        C.c_float(sigma_pixel), C.c_float(sigma_depth), C.byref(self.cam)), "kf_prefetch_frame")

    def downsample(self, model):
        if model:
            _chk(self.lib.kf_downsample_model_vertices(self.h), "kf_downsample_model_vertices")
            _chk(self.lib.kf_downsample_model_normals(self.h), "kf_downsample_model_normals")
        else:
            _chk(self.lib.kf_downsample_new_vertices(self.h), "kf_downsample_new_vertices")
            _chk(self.lib.kf_downsample_new_normals(self.h), "kf_downsample_new_normals")

    def icp_system(self, level, cur, last_inv, cam, dist, sin):
        _chk(self.lib.kf_cal_point_to_plane_solver_params(self.h, level, C.byref(Mat44.of(cur)), C.byref(Mat44.of(last_inv)),
                                                          C.byref(cam), C.c_float(dist), C.c_float(sin)), "kf_cal_point_to_plane_solver_params")
        return self.read_solver_params()

    def sdf_system(self, cur):
        _chk(self.lib.kf_cal_sdf_solver_params(self.h, C.byref(self.cam), C.byref(Mat44.of(cur))), "kf_cal_sdf_solver_params")
        return self.read_solver_params()

    def read_solver_params(self):
        out = np.zeros(27, np.float32)
        _chk(self.lib.kf_read_solver_params(self.h, _p(out)), "kf_read_solver_params")
        return out

    # ---- device-resident tracking ----
    def set_pose(self, pose):
        _chk(self.lib.kf_set_pose(self.h, C.byref(Mat44.of(pose))), "kf_set_pose")

    def icp_track(self, frame_id, dist, sin, dist_shake, angle_shake):
        p = IcpParams(self.levels, sin, dist, dist_shake, angle_shake)
        _chk(self.lib.kf_icp_track(self.h, frame_id, C.byref(p), C.byref(self.cam)), "kf_icp_track")

    def icp_partition_track(self, frame_id, dist, sin, dist_shake, angle_shake, part, parts, dev_sums_ptr, all_reduce):
        """ICP with the pixels split over `parts` ranks; `all_reduce()` must sum the 32-float buffer at dev_sums_ptr over the ranks."""
        p = IcpParams(self.levels, sin, dist, dist_shake, angle_shake)
        _chk(self.lib.kf_icp_partition_begin(self.h, frame_id), "kf_icp_partition_begin")
        if frame_id == 0:
            return
        for step in range(self.lib.kf_icp_partition_steps(self.h)):
            _chk(self.lib.kf_icp_partition_step(self.h, step, C.byref(p), C.byref(self.cam), part, parts, C.c_void_p(dev_sums_ptr)),
                 "kf_icp_partition_step")
            all_reduce()
        _chk(self.lib.kf_icp_partition_finish(self.h, C.byref(p), C.c_void_p(dev_sums_ptr)), "kf_icp_partition_finish")

    def sdf_partition_track(self, frame_id, max_iter, dist_shake, angle_shake, dev_sums_ptr, all_reduce):
        """CameraPoseFinderSDF on a z-slab: this context sums the pixels whose world point it owns; `all_reduce()` must sum the 32-float
        buffer at dev_sums_ptr over the ranks (called max_iter times on every rank, whatever the convergence)."""
        p = SdfTrackerParams(max_iter, dist_shake, angle_shake)
        _chk(self.lib.kf_sdf_partition_begin(self.h, frame_id), "kf_sdf_partition_begin")
        if frame_id == 0:
            return
        for step in range(max_iter):
            _chk(self.lib.kf_sdf_partition_step(self.h, step, C.byref(p), C.byref(self.cam), C.c_void_p(dev_sums_ptr)), "kf_sdf_partition_step")
            all_reduce()
        _chk(self.lib.kf_sdf_partition_finish(self.h, C.byref(p), C.byref(self.cam), C.c_void_p(dev_sums_ptr)), "kf_sdf_partition_finish")

    def sdf_track(self, frame_id, max_iter, dist_shake, angle_shake):
        p = SdfTrackerParams(max_iter, dist_shake, angle_shake)
        _chk(self.lib.kf_sdf_track(self.h, frame_id, C.byref(p), C.byref(self.cam)), "kf_sdf_track")

    def inject_track_stall(self, launches=1):
        """fault injection: the next `launches` persistent-loop launches run with one workgroup playing dead (the frame is finished solo)"""
        _chk(self.lib.kf_inject_track_stall(self.h, int(launches)), "kf_inject_track_stall")

    def cull_tail_counts(self):
        """(consumed, undone): culls that ran as the tail of a tracking launch and were used by / discarded before the fusion pass"""
        a, b = C.c_uint32(), C.c_uint32()
        _chk(self.lib.kf_cull_tail_counts(self.h, C.byref(a), C.byref(b)), "kf_cull_tail_counts")
        return a.value, b.value

    def track_result(self):
        r = TrackResult()
        _chk(self.lib.kf_read_track_result(self.h, C.byref(r)), "kf_read_track_result")
        self.last_form = r.launch_form       # 1: persistent device loop, 2: one launch per Gauss-Newton step, 3: loop finished by one workgroup after a time-out (same pose bits), 0: none
        return bool(r.tracked), r.pose.numpy(), r.status, r.iterations

    # ---- volume ----
    def integrate(self, pose, sdf_trunc, max_dist, has_color=False, angle_weight=False):
        ip = IntegrateParams(sdf_trunc, max_dist)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_integrate_volume(self.h, int(has_color), int(angle_weight), tp, C.byref(ip), C.byref(self.cam),
                                          C.byref(self.rgb_cam)), "kf_integrate_volume")

    def set_defer(self, mode):
        """deferred free-space weights: 1 on, 0 off (the plain fusion kernel on every frame), -1 follow the environment"""
        _chk(self.lib.kf_set_defer(self.h, int(mode)), "kf_set_defer")

    def raycast(self, pose, inc, near, far, has_color=False):
        rp = RaycastParams(inc)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_raycast_volume(self.h, int(has_color), tp, C.byref(rp), C.byref(self.cam), C.c_float(near), C.c_float(far)),
             "kf_raycast_volume")

    def raycast_slab(self, pose, inc, near, far, dev_t, dev_v, dev_n):
        rp = RaycastParams(inc)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_raycast_volume_slab(self.h, 0, tp, C.byref(rp), C.byref(self.cam), C.c_float(near), C.c_float(far),
                                             C.c_void_p(dev_t), C.c_void_p(dev_v), C.c_void_p(dev_n)), "kf_raycast_volume_slab")

    def slab_mask_candidates(self, dev_t, dev_tmin, dev_v, dev_n):
        _chk(self.lib.kf_slab_mask_candidates(self.h, C.c_void_p(dev_t), C.c_void_p(dev_tmin), C.c_void_p(dev_v), C.c_void_p(dev_n)),
             "kf_slab_mask_candidates")

    def raycast_slab_cross(self, pose, inc, near, far, dev_ta):
        """every ray's first crossing in the owned layers as one 64-bit word per pixel (crossing parameter << 32 | vertex parameter alpha)"""
        rp = RaycastParams(inc)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_raycast_volume_slab_cross(self.h, tp, C.byref(rp), C.byref(self.cam), C.c_float(near), C.c_float(far), C.c_void_p(dev_ta)),
             "kf_raycast_volume_slab_cross")

    def slab_ray_normals(self, pose, inc, near, far, dev_ta_min, dev_cand):
        """after the MIN all-reduce of the words: (normal, 1) for the winners whose vertex this context owns, zeros elsewhere"""
        rp = RaycastParams(inc)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_slab_ray_normals(self.h, tp, C.byref(rp), C.byref(self.cam), C.c_float(near), C.c_float(far), C.c_void_p(dev_ta_min),
                                          C.c_void_p(dev_cand)), "kf_slab_ray_normals")

    def raycast_slab_cross_spec(self, pose, inc, near, far, dev_ta, dev_ta_own, dev_spec):
        """raycast_slab_cross + a second copy of the words and the speculative normals of this context's own crossings (kf_raycast_volume_slab_cross_spec)"""
        rp = RaycastParams(inc)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_raycast_volume_slab_cross_spec(self.h, tp, C.byref(rp), C.byref(self.cam), C.c_float(near), C.c_float(far), C.c_void_p(dev_ta),
                                                        C.c_void_p(dev_ta_own), C.c_void_p(dev_spec)), "kf_raycast_volume_slab_cross_spec")

    def slab_ray_normals_spec(self, pose, inc, near, far, dev_ta_min, dev_ta_own, dev_spec, dev_cand):
        """slab_ray_normals that copies the speculative normal where this context's own crossing won and evaluates the rest"""
        rp = RaycastParams(inc)
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_slab_ray_normals_spec(self.h, tp, C.byref(rp), C.byref(self.cam), C.c_float(near), C.c_float(far), C.c_void_p(dev_ta_min),
                                               C.c_void_p(dev_ta_own), C.c_void_p(dev_spec), C.c_void_p(dev_cand)), "kf_slab_ray_normals_spec")

    def set_model_maps_rays(self, pose, dev_ta_min, dev_cand):
        tp = C.byref(Mat44.of(pose)) if pose is not None else None
        _chk(self.lib.kf_set_model_maps_rays(self.h, tp, C.byref(self.cam), C.c_void_p(dev_ta_min), C.c_void_p(dev_cand)), "kf_set_model_maps_rays")

    def set_model_maps_device(self, dev_v, dev_n):
        _chk(self.lib.kf_set_model_maps_device(self.h, C.c_void_p(dev_v), C.c_void_p(dev_n)), "kf_set_model_maps_device")

    def set_stream(self, hip_stream):
        _chk(self.lib.kf_set_stream(self.h, C.c_void_p(hip_stream)), "kf_set_stream")

    def marching_cubes(self, thr, has_color=False):
        _chk(self.lib.kf_marching_cubes(self.h, int(has_color), C.c_float(thr)), "kf_marching_cubes")

    def clear_triangles(self):
        _chk(self.lib.kf_clear_triangles(self.h), "kf_clear_triangles")

    def triangles(self):
        n = C.c_uint32()
        _chk(self.lib.kf_triangle_count(self.h, C.byref(n)), "kf_triangle_count")
        out = np.zeros(n.value, dtype=TRI_DTYPE)
        if n.value:
            _chk(self.lib.kf_read_triangles(self.h, _p(out), 0, n.value), "kf_read_triangles")
        return out

    def download_volume(self, z0=None, z1=None, color=False):
        z0 = self.stored[0] if z0 is None else z0
        z1 = self.stored[1] if z1 is None else z1
        shape = (z1 - z0, self.res, self.res)
        t, w = np.empty(shape, np.float32), np.empty(shape, np.float32)
        c = np.empty(shape + (3,), np.uint8) if color else None
        _chk(self.lib.kf_download_volume(self.h, z0, z1, _p(t), _p(w), _p(c) if color else None), "kf_download_volume")
        return (t, w, c) if color else (t, w)

    def upload_volume(self, tsdf, weight, color=None, z0=None):
        z0 = self.stored[0] if z0 is None else z0
        tsdf, weight = np.ascontiguousarray(tsdf, np.float32), np.ascontiguousarray(weight, np.float32)
        z1 = z0 + tsdf.shape[0]
        cc = np.ascontiguousarray(color, np.uint8) if color is not None else None
        _chk(self.lib.kf_upload_volume(self.h, z0, z1, _p(tsdf), _p(weight), _p(cc) if cc is not None else None), "kf_upload_volume")

    def download_volume_device(self, z0, z1, dev_tsdf, dev_weight):
        _chk(self.lib.kf_download_volume_device(self.h, z0, z1, C.c_void_p(dev_tsdf), C.c_void_p(dev_weight), None), "kf_download_volume_device")

    def upload_volume_device(self, z0, z1, dev_tsdf, dev_weight):
        _chk(self.lib.kf_upload_volume_device(self.h, z0, z1, C.c_void_p(dev_tsdf), C.c_void_p(dev_weight), None), "kf_upload_volume_device")

    def resize_slab(self, z0, z1, halo):
        """own [z0, z1) (+ halo) from now on: layers stored before and after keep their voxels, new ones read as never observed"""
        _chk(self.lib.kf_resize_slab(self.h, z0, z1, halo), "kf_resize_slab")
        z0s, z1s = C.c_uint32(), C.c_uint32()
        _chk(self.lib.kf_stored_z_range(self.h, C.byref(z0s), C.byref(z1s)), "kf_stored_z_range")
        self.stored, self.owned = (z0s.value, z1s.value), (z0, z1)

    def count_layer_work(self, frames):
        _chk(self.lib.kf_count_layer_work(self.h, int(frames)), "kf_count_layer_work")

    def read_layer_work(self, reset=True):
        out = np.zeros(self.res // 8, np.uint64)
        _chk(self.lib.kf_read_layer_work(self.h, _p(out), int(reset)), "kf_read_layer_work")
        return out

    def reset_volume(self):
        _chk(self.lib.kf_reset_volume(self.h), "kf_reset_volume")

    def stats(self, observed=True):
        """observed=False: kf_get_fusion_counters -- the update counters only (weight_gt0 = 0), no sweep, no effect on the observed-voxel count's bookkeeping"""
        s = VolumeStats()
        if not observed:
            _chk(self.lib.kf_get_fusion_counters(self.h, C.byref(s)), "kf_get_fusion_counters")
            return dict(updated_last=s.updated_last, weight_gt0=0, bricks_active=s.bricks_active, bricks_total=s.bricks_total,
                        updated_total=s.updated_total, frames_fused=s.frames_fused, frames_lost=s.frames_lost)
        _chk(self.lib.kf_get_volume_stats(self.h, C.byref(s)), "kf_get_volume_stats")
        if os.environ.get("KF_STATS_CROSSCHECK") == "1":       # tests/conftest.py: every stats() call checks the running count against a sweep of the volume
            swept = self.count_observed_voxels()
            if swept != s.weight_gt0:
                raise AssertionError("kf_get_volume_stats: running count of observed voxels %d != swept count %d" % (s.weight_gt0, swept))
        return dict(updated_last=s.updated_last, weight_gt0=s.weight_gt0, bricks_active=s.bricks_active, bricks_total=s.bricks_total,
                    updated_total=s.updated_total, frames_fused=s.frames_fused, frames_lost=s.frames_lost)

    def count_observed_voxels(self):
        """voxels of the owned layers with weight > 0, by a sweep of the volume (kf_get_volume_stats keeps the same number as a running count)"""
        n = C.c_uint64()
        _chk(self.lib.kf_count_observed_voxels(self.h, C.byref(n)), "kf_count_observed_voxels")
        return int(n.value)

    def stage_timers(self, mask):
        _chk(self.lib.kf_stage_timers(self.h, int(mask)), "kf_stage_timers")

    def read_stage_ms(self):
        ms = np.zeros(8, np.float32)
        cnt = np.zeros(8, np.uint32)
        _chk(self.lib.kf_read_stage_ms(self.h, _p(ms), _p(cnt)), "kf_read_stage_ms")
        return ms, cnt

    def work_counters(self):
        """(raycast reference samples, rays evaluated, bricks read by marching cubes, triangles) since stage_timers(mask | 1 << 16)"""
        out = (C.c_uint64 * 4)()
        _chk(self.lib.kf_read_work_counters(self.h, out), "kf_read_work_counters")
        return tuple(int(v) for v in out)

    def selftest_div(self, n, seed, mode):
        m = C.c_uint32(0)
        _chk(self.lib.kf_selftest_div(self.h, n, seed, mode, C.byref(m)), "kf_selftest_div")
        return m.value

    def sync(self):
        _chk(self.lib.kf_synchronize(self.h), "kf_synchronize")

    @property
    def stream(self):
        return self.lib.kf_stream(self.h)
