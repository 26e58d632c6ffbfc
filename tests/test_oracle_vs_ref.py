"""Pin the oracle's helper arithmetic against the REFERENCE's own __host__ __device__ headers.

oracle/_ref/libkfref.so is built (oracle/Makefile) from /root/reference/src/cuda/{tsdfVolume,Mat,DepthCamera,
cuda_declar}.h as they lie.  Everything here is bit-exact.  Skipped when the .so is absent.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libkfref.so not built (needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rand_pose(rng, scale=1.0):
    a = rng.normal(size=3) * 0.3
    th = np.linalg.norm(a)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = rng.normal(size=3) * scale
    return m.astype(np.float32)


def test_struct_sizes():
    r = O.ref()
    assert r.ref_sizeof_voxel() == 12 == O.VOXEL_DTYPE.itemsize
    assert r.ref_sizeof_camera_params() == 24 == C.sizeof(O.Cam)


def test_mat44_inverse_mul_vec():
    rng = np.random.default_rng(1)
    r = O.ref()
    for _ in range(200):
        a, b = rand_pose(rng, 2.0), rand_pose(rng, 2.0)
        out = np.zeros(16, np.float32)
        r.ref_mat44_inverse(O.fp(O.m16(a)), O.fp(out))
        assert np.array_equal(bits(out), bits(O.mat44_inverse(a).reshape(16)))
        r.ref_mat44_mul(O.fp(O.m16(a)), O.fp(O.m16(b)), O.fp(out))
        assert np.array_equal(bits(out), bits(O.mat44_mul(a, b).reshape(16)))


def _vol_pair(rng, res=16, size=1.5, fill=0.8):
    r = O.ref()
    h = C.c_void_p(r.ref_vol_create(res, C.c_float(size), C.c_float(128.0)))
    vol = O.OVolume(res, size, 128.0)
    n = res ** 3
    vol.vox["tsdf"] = rng.uniform(-1, 1, n).astype(np.float32)
    vol.vox["weight"] = np.where(rng.uniform(size=n) < fill, rng.integers(1, 128, n), 0).astype(np.float32)
    vol.vox["color"] = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    for z in range(res):
        for y in range(res):
            for x in range(res):
                q = vol.vox[(z * res + y) * res + x]
                col = (C.c_ubyte * 3)(*[int(c) for c in q["color"]])
                r.ref_vol_set(h, x, y, z, C.c_float(q["tsdf"]), C.c_float(q["weight"]), col)
    return r, h, vol


def test_voxel_world_maps_and_interpolation():
    rng = np.random.default_rng(2)
    r, h, vol = _vol_pair(rng, fill=0.97)
    res, size = vol.res, vol.size
    n_ok = 0
    for _ in range(4000):
        p = rng.uniform(-0.1, size + 0.1, 3).astype(np.float32)
        d_ref = C.c_float(0)
        ok_ref = r.ref_vol_interp_sdf(h, O.fp(p), C.byref(d_ref))
        ok, d = O.interpolate_sdf(vol, p)
        assert bool(ok_ref) == ok
        if ok:
            n_ok += 1
            assert np.float32(d_ref.value).view(np.uint32) == np.float32(d).view(np.uint32)
        t, w = C.c_float(0), C.c_float(0)
        r.ref_vol_nearest(h, O.fp(p), C.byref(t), C.byref(w))
        g = np.clip((p * np.float32(res) / np.float32(size)).astype(np.int32), 0, res - 1)
        assert t.value == vol.tsdf[g[2], g[1], g[0]] and w.value == vol.weight[g[2], g[1], g[0]]
    assert n_ok > 500
    r.ref_vol_destroy(h)


def test_update_voxel_running_average():
    """tsdfVolume.h:57-75 through okf_integrate: one depth pixel per voxel column is hard to arrange, so
    drive both sides with the same per-voxel sequence via a 1-voxel-deep synthetic integrate instead:
    compare the reference updateVoxel chain with the oracle's closed form on random sequences."""
    rng = np.random.default_rng(3)
    r = O.ref()
    h = C.c_void_p(r.ref_vol_create(4, C.c_float(1.0), C.c_float(8.0)))
    col = (C.c_ubyte * 3)(10, 200, 31)
    for trial in range(50):
        x, y, z = [int(v) for v in rng.integers(0, 4, 3)]
        t_old, w_old = C.c_float(0), C.c_float(0)
        c_old = (C.c_ubyte * 3)()
        for _ in range(12):
            r.ref_vol_get(h, x, y, z, C.byref(t_old), C.byref(w_old), c_old)
            tsdf = np.float32(rng.uniform(-1, 1))
            r.ref_vol_update(h, x, y, z, C.c_float(tsdf), C.c_float(1.0), col, C.c_float(2.0))
            t_new, w_new = C.c_float(0), C.c_float(0)
            c_new = (C.c_ubyte * 3)()
            r.ref_vol_get(h, x, y, z, C.byref(t_new), C.byref(w_new), c_new)
            ow, ot = np.float32(w_old.value), np.float32(t_old.value)
            exp_w = min(np.float32(ow + np.float32(1)), np.float32(8.0))
            exp_t = np.float32(np.float32(ot * ow + tsdf * np.float32(1)) / np.float32(ow + np.float32(1)))
            assert w_new.value == exp_w and np.float32(t_new.value).view(np.uint32) == exp_t.view(np.uint32)
            for k in range(3):
                oc = np.float32(c_old[k])
                nc = min(np.float32(255.0), np.float32(np.float32(oc * ow + np.float32(col[k]) * np.float32(2.0)) / np.float32(ow + np.float32(2.0))))
                assert c_new[k] == int(nc)
    r.ref_vol_destroy(h)


def test_integrate_matches_reference_helpers():
    """Run the oracle's integrate on a small volume and replay every voxel through the reference's
    voxelPosToWorld / Mat44 / projectSkeletonToScreen / updateVoxel: identical tsdf+weight bits."""
    rng = np.random.default_rng(4)
    r = O.ref()
    res, size = 24, 1.2
    cam = O.Cam.make(64, 48, 31.5, 23.5, 52.5, 52.5)
    depth = rng.uniform(0.35, 1.6, (48, 64)).astype(np.float32)
    depth[rng.uniform(size=depth.shape) < 0.1] = 0
    nrm = np.zeros((48, 64, 4), np.float32)
    pose = rand_pose(rng, 0.0)
    pose[:3, 3] = [0.6, 0.55, -0.3]
    vol = O.OVolume(res, size, 128.0)
    h = C.c_void_p(r.ref_vol_create(res, C.c_float(size), C.c_float(128.0)))
    col = (C.c_ubyte * 3)(0, 0, 0)
    n_ref = 0
    for rep in range(2):
        n_upd = O.integrate(vol, depth, nrm, None, False, False, pose, 0.05, 2.0, cam, cam)
        tinv = np.zeros(16, np.float32)
        r.ref_mat44_inverse(O.fp(O.m16(pose)), O.fp(tinv))
        n_ref = 0
        for z in range(res):
            for y in range(res):
                for x in range(res):
                    w = np.zeros(4, np.float32)
                    r.ref_vol_voxel_to_world(h, x, y, z, O.fp(w))
                    w[3] = 1.0
                    pf = np.zeros(4, np.float32)
                    r.ref_mat44_vec(O.fp(tinv), O.fp(w), O.fp(pf))
                    if pf[2] <= 0:
                        continue
                    sp = np.zeros(2, np.int32)
                    r.ref_project_to_screen(O.fp(pf), C.byref(cam), O.fp(sp))
                    if sp[0] >= 63 or sp[1] >= 47 or sp[0] < 1 or sp[1] < 1:
                        continue
                    d = depth[sp[1], sp[0]]
                    if d == 0 or not d < np.float32(2.0):
                        continue
                    sdf = np.float32(d - pf[2])
                    if sdf > np.float32(-0.05):
                        tsdf = min(np.float32(1.0), np.float32(sdf / np.float32(0.05)))
                        r.ref_vol_update(h, x, y, z, C.c_float(tsdf), C.c_float(1.0), col, C.c_float(2.0))
                        n_ref += 1
        assert n_ref == n_upd and n_upd > 500
    t, w = C.c_float(0), C.c_float(0)
    c = (C.c_ubyte * 3)()
    for z in range(res):
        for y in range(res):
            for x in range(res):
                r.ref_vol_get(h, x, y, z, C.byref(t), C.byref(w), c)
                assert np.float32(t.value).view(np.uint32) == vol.tsdf[z, y, x].view(np.uint32)
                assert w.value == vol.weight[z, y, x]
    r.ref_vol_destroy(h)


def test_vector_helpers():
    rng = np.random.default_rng(5)
    r = O.ref()
    cam = O.Cam.make(640, 480, 319.5, 239.5, 525.0, 525.0)
    d = np.ones((480, 640), np.float32)
    d[:] = rng.uniform(0.4, 3.0, d.shape)
    v = O.depth_to_vertices(d, cam)
    n = O.vertices_to_normals(v)
    out = np.zeros(3, np.float32)
    for _ in range(300):
        x, y = int(rng.integers(1, 639)), int(rng.integers(1, 479))
        r.ref_depth_to_skeleton(x, y, C.c_float(d[y, x]), C.byref(cam), O.fp(out))
        assert np.array_equal(bits(out), bits(v[y, x, :3]))
        up = (v[y + 1, x, :3] - v[y - 1, x, :3]).astype(np.float32)
        rt = (v[y, x + 1, :3] - v[y, x - 1, :3]).astype(np.float32)
        cr = np.zeros(3, np.float32)
        r.ref_cross(O.fp(up), O.fp(rt), O.fp(cr))
        r.ref_normalize(O.fp(cr), O.fp(out))
        assert np.array_equal(bits(out), bits(n[y, x, :3]))


def test_color_interpolation_world_to_voxel_and_norm():
    """tsdfVolume.h:123-148 interpolateColor (validity + the truncating float->uchar result), :50-56 worldPosToVoxel and
    cuda_declar.h norm(): the oracle's restatements against the reference's own headers, bit for bit."""
    rng = np.random.default_rng(9)
    r, h, vol = _vol_pair(rng, fill=0.97)
    size = vol.size
    n_ok = 0
    for _ in range(4000):
        p = rng.uniform(-0.1, size + 0.1, 3).astype(np.float32)
        c_ref = (C.c_ubyte * 3)(0, 0, 0)
        ok_ref = r.ref_vol_interp_color(h, O.fp(p), c_ref)
        ok, c = O.interpolate_color(vol, p)
        assert bool(ok_ref) == ok
        if ok:
            n_ok += 1
            assert list(c_ref) == [int(x) for x in c]
        g_ref = np.zeros(3, np.int32)
        r.ref_vol_world_to_voxel(h, O.fp(p), O.fp(g_ref))
        assert np.array_equal(g_ref, O.world_to_voxel(vol, p))
        v = (rng.standard_normal(3) * rng.choice([1e-6, 1.0, 1e4])).astype(np.float32)
        assert np.float32(r.ref_norm(O.fp(v))).view(np.uint32) == np.float32(O.norm(v)).view(np.uint32)
    assert n_ok > 500
    r.ref_vol_destroy(h)


def _unpack_mc_inc(path):
    """csrc/mc_tables.inc / oracle/mc_tables.inc: 256 64-bit words, nibble i = edge id of triTable[c][i], 0xF = -1 (tools/pack_mc_tables.py)."""
    import re
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-fA-F]{16})ull", open(path).read())]
    assert len(words) == 256
    tri = np.full((256, 16), -1, np.int32)
    edge = np.zeros(256, np.int32)
    for c, w in enumerate(words):
        for i in range(16):
            nib = (w >> (4 * i)) & 0xF
            if nib != 0xF:
                tri[c, i] = nib
                edge[c] |= 1 << nib
    return edge, tri


# sha256 of the reference's tables as its own header compiles them (int32 little endian: edgeTable[256], triTable[256][16]), recorded from
# oracle/_ref/libkfref.so::ref_mc_tables so that the pin also holds where /root/reference is absent (test_mc_tables_hash_without_reference)
MC_EDGE_SHA256 = "ffc58719f11be7a8b34988740a15dcd043dc314a3e2fe01e917fd796001815b9"
MC_TRI_SHA256 = "85e6eb7486ad0101a95aaf3b332a15ba6e5d187a24d31c5eb77874d3aa45d996"


def test_marching_cubes_tables_equal_the_reference_header():
    """edgeTable / triTable of src/cuda/marchingcube_table.h:19,56, compiled as they lie, against the packed tables the HIP kernels
    (hybkinectfu_amd/csrc/mc_tables.inc) and the oracle (oracle/mc_tables.inc) carry: element for element."""
    import hashlib
    import os
    r = O.ref()
    edge = np.zeros(256, np.int32)
    tri = np.zeros(256 * 16, np.int32)
    r.ref_mc_tables(O.fp(edge), O.fp(tri))
    tri = tri.reshape(256, 16)
    assert hashlib.sha256(edge.tobytes()).hexdigest() == MC_EDGE_SHA256 and hashlib.sha256(tri.tobytes()).hexdigest() == MC_TRI_SHA256
    for rel in ("hybkinectfu_amd/csrc/mc_tables.inc", "oracle/mc_tables.inc"):
        e, t = _unpack_mc_inc(os.path.join(O.ROOT, rel))
        assert np.array_equal(t, tri), rel
        assert np.array_equal(e, edge), rel          # the 12-bit edge mask of a case is exactly the set of edges its triangles use
