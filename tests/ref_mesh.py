"""ctypes binding of oracle/_ref/libkfrefmesh.so: the REFERENCE's own ml::MeshData<float> / ml::MeshIO<float>, compiled from
/root/reference/src where it lies (oracle/ref_mesh_harness.cpp).  Exists only in the build container; test infrastructure."""
import ctypes as C
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libkfrefmesh.so")


def available():
    return os.path.exists(PATH)


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        _lib.ref_mesh_build.restype = C.c_void_p
    return _lib


def process(triangles, with_color=False, save_as=()):
    """MeshGeneratorMarchingcube::saveMesh's sequence on a triangle soup through the reference's own classes; every name in
    save_as is written with ml::MeshIOf::saveToFile."""
    lib = _load()
    tris = np.ascontiguousarray(triangles)
    assert tris.dtype.itemsize == 72
    h = C.c_void_p(lib.ref_mesh_build(tris.ctypes.data_as(C.c_void_p), len(tris), int(with_color)))
    try:
        nv, nf, nc = lib.ref_mesh_vertex_count(h), lib.ref_mesh_face_count(h), lib.ref_mesh_color_count(h)
        v = np.empty((nv, 3), np.float32)
        n = np.empty((nv, 3), np.float32)
        c = np.empty((nc, 4), np.float32)
        f = np.empty((nf, 3), np.uint32)
        odd = lib.ref_mesh_read(h, v.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p) if nc else None,
                                f.ctypes.data_as(C.c_void_p))
        assert odd == 0
        for name in save_as:
            lib.ref_mesh_save(h, name.encode())
    finally:
        lib.ref_mesh_destroy(h)
    return dict(vertices=v, normals=n, colors=c, faces=f)
