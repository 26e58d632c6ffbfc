"""ctypes binding of libhybkf_host.so: the C++ host classes (HybKinectfu, CameraPoseFinder*, MeshGeneratorMarchingcube)
driven the way src/MainController.cpp drives the reference.  One application per process (the reference uses singletons)."""
import ctypes as C
import os

import numpy as np

from . import lib as K

HOST_LIB = os.path.join(K.PKG_DIR, "libhybkf_host.so")
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
                 trunc_max=0.0, device=0, slab=(0, 0), halo=0):
        self.h = load()
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
