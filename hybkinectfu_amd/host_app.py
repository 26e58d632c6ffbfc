"""ctypes binding of libhybkf_host.so: the C++ host classes (HybKinectfu, CameraPoseFinder*, MeshGeneratorMarchingcube)
driven the way src/MainController.cpp drives the reference.  One application per process (the reference uses singletons)."""
import ctypes as C
import os

import numpy as np

from . import lib as K

HOST_LIB = os.environ.get("KF_HOST_LIB") or os.path.join(K.PKG_DIR, "libhybkf_host.so")     # KF_HOST_LIB: the sanitizer build (CPU tests)
_h = None


def load():
    global _h
    if _h is None:
        K.load()
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("libhybkf_host.so is missing: run __graft_entry__.build()")
        _h = C.CDLL(HOST_LIB)
        _h.hkf_app_ctx.restype = C.c_void_p
    return _h


class App:
    def __init__(self, res, size, cam, sdf_tracker=False, host_loop=False, max_triangles=0, sdf_trunc=0.0, integrate_dist=0.0,
                 trunc_max=0.0, device=0, slab=(0, 0), halo=0, dataset_dir="", traj_read="", traj_write="", use_rgb=False):
        """dataset_dir: TUM RGB-D directory (with trailing slash) read by process_dataset_frame; traj_read: trajectory file used
        INSTEAD of a tracker (CameraPoseFinderFromFile); traj_write: record the tracked poses there (TrajectoryRecorder)."""
        self.h = load()
        self.h.hkf_app_configure_io(dataset_dir.encode(), traj_read.encode(), traj_write.encode(), int(use_rgb))
        st = self.h.hkf_app_init(res, C.c_float(size), cam[0], cam[1], C.c_float(cam[2]), C.c_float(cam[3]), C.c_float(cam[4]), C.c_float(cam[5]),
                                 int(sdf_tracker), int(host_loop), max_triangles, C.c_float(sdf_trunc), C.c_float(integrate_dist),
                                 C.c_float(trunc_max), device, slab[0], slab[1], halo)
        if st:
            raise K.KfError("hkf_app_init failed: %d" % st)
        self.cam = cam

    def process_frame(self, mm, frame_id, stamp=0.0):
        mm = np.ascontiguousarray(mm, np.uint16)
        r = self.h.hkf_app_process_frame(mm.ctypes.data_as(C.c_void_p), 0, frame_id, C.c_double(stamp))
        if r < 0:
            raise K.KfError("processNewFrame failed: %d" % r)
        return bool(r)

    def process_dataset_frame(self, frame_id):
        """Next frame of the dataset directory through processNewFrame.  Returns (tracked, depth time stamp) or None at the end."""
        stamp = C.c_double(0.0)
        r = self.h.hkf_app_process_dataset_frame(frame_id, C.byref(stamp))
        if r == -3:
            return None
        if r < 0:
            raise K.KfError("dataset frame failed: %d" % r)
        return bool(r), stamp.value

    def enqueue_frame_device(self, dev_ptr, frame_id):
        r = self.h.hkf_app_enqueue_frame(C.c_void_p(dev_ptr), 1, frame_id)
        if r < 0:
            raise K.KfError("enqueueFrame failed: %d" % r)

    def pose(self):
        out = np.zeros(16, np.float32)
        tracked = self.h.hkf_app_get_pose(out.ctypes.data_as(C.c_void_p))
        return bool(tracked), out.reshape(4, 4)

    def ctx_handle(self):
        return C.c_void_p(self.h.hkf_app_ctx())

    def generate_mesh(self):
        return self.h.hkf_app_generate_mesh()

    def save_mesh(self, filename):
        nv, nf = C.c_uint32(), C.c_uint32()
        ok = self.h.hkf_app_save_mesh(filename.encode(), C.byref(nv), C.byref(nf))
        return bool(ok), nv.value, nf.value

    def close(self):
        self.h.hkf_app_shutdown()


# ---- GPU-free helpers of the dataset / trajectory code -------------------------------------------------------------------------
def dataset_read(directory, cols, rows, n, with_color=False):
    h = load()
    depth = np.zeros((n, rows, cols), np.uint16)
    bgr = np.zeros((n, rows, cols, 3), np.uint8) if with_color else None
    ds, cs = np.zeros(n, np.float64), np.zeros(n, np.float64)
    got = h.hkf_dataset_read(directory.encode(), cols, rows, int(with_color), n, depth.ctypes.data_as(C.c_void_p),
                             bgr.ctypes.data_as(C.c_void_p) if with_color else None, ds.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p))
    if got < 0:
        raise K.KfError("hkf_dataset_read failed: %d" % got)
    return depth[:got], (bgr[:got] if with_color else None), ds[:got], cs[:got]


def png_read(path):
    h = load()
    h.hkf_png_read.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    w, hh, ch, bits = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    if not h.hkf_png_read(path.encode(), C.byref(w), C.byref(hh), C.byref(ch), C.byref(bits), None, 0):
        return None
    out = np.zeros((hh.value, w.value, ch.value), np.uint16 if bits.value == 16 else np.uint8)
    h.hkf_png_read(path.encode(), C.byref(w), C.byref(hh), C.byref(ch), C.byref(bits), out.ctypes.data_as(C.c_void_p), out.nbytes)
    return out


def pyrdown16(img):
    h = load()
    img = np.ascontiguousarray(img, np.uint16)
    out = np.zeros(((img.shape[0] + 1) // 2, (img.shape[1] + 1) // 2), np.uint16)
    h.hkf_pyrdown16(img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], out.ctypes.data_as(C.c_void_p))
    return out


def quat_from_pose(pose):
    h = load()
    p = np.ascontiguousarray(pose, np.float32).reshape(16)
    q = np.zeros(4, np.float32)
    h.hkf_quat_from_pose(p.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p))
    return q


def pose_from_quat(t, q_xyzw):
    h = load()
    t = np.ascontiguousarray(t, np.float32); q = np.ascontiguousarray(q_xyzw, np.float32)
    out = np.zeros(16, np.float32)
    h.hkf_pose_from_quat(t.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out.reshape(4, 4)


def trajectory_write(path, poses, stamps):
    h = load()
    p = np.ascontiguousarray(poses, np.float32).reshape(-1, 16); s = np.ascontiguousarray(stamps, np.float64)
    return h.hkf_trajectory_write(path.encode(), p.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), len(s))


def table_nearest(path, header_lines, targets):
    h = load()
    t = np.ascontiguousarray(targets, np.float64); out = np.zeros(len(t), np.float64)
    n = h.hkf_table_nearest(path.encode(), header_lines, t.ctypes.data_as(C.c_void_p), len(t), out.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise K.KfError("cannot read " + path)
    return out


# ---- mesh post-processing (MeshGeneratorMarchingcube::saveMesh's weld / dedupe / normals / writers) ------------------------------
def _mesh_read(which):
    h = load()
    nv, nf, nc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if h.hkf_mesh_sizes(which, C.byref(nv), C.byref(nf), C.byref(nc)) != 0:
        raise K.KfError("no mesh")
    v = np.empty((nv.value, 3), np.float32)
    n = np.empty((nv.value, 3), np.float32)
    c = np.empty((nc.value, 4), np.float32)
    f = np.empty((nf.value, 3), np.uint32)
    h.hkf_mesh_read(which, v.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p) if nc.value else None,
                    f.ctypes.data_as(C.c_void_p))
    return dict(vertices=v, normals=n, colors=c, faces=f)


def mesh_from_soup(triangles, with_color=False):
    """GPU-free: weld / dedupe / normals of a triangle soup (numpy array of lib.TRI_DTYPE), as saveMesh does after the copy."""
    h = load()
    tris = np.ascontiguousarray(triangles)
    assert tris.dtype.itemsize == 72
    h.hkf_mesh_from_soup(tris.ctypes.data_as(C.c_void_p), len(tris), int(with_color), None, None)
    return _mesh_read(0)


def mesh_save(which, filename):
    """which: 0 = the mesh of mesh_from_soup, 1 = the application's mesh (after App.save_mesh)"""
    return bool(load().hkf_mesh_save(which, filename.encode()))


def app_mesh():
    return _mesh_read(1)
