"""GPU parity: every HIP stage, called through the C ABI (libhybkf.so), against the CPU oracle on identical inputs.

Bit-exact: depth conversion, gate, vertices, normals, pyramids, TSDF planes + update counts, raycast maps, marching-cubes
triangle sequence.  Tolerance (stated per test): bilateral (device __expf), the 27 fp32 ICP/SDF sums, tracked poses.
"""
import numpy as np
import pytest
from conftest import default_forms

import oracle_lib as O
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import scene as S

pytestmark = pytest.mark.gpu

P = S.STOCK


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def small_cam():
    return (64, 48, 31.5, 23.5, 52.5, 52.5)


def mid_cam():
    return (160, 120, 79.5, 59.5, 131.25, 131.25)


def ragged_cam():
    """Not a multiple of any tile the kernels use (64x4 bilateral, 32x16 raycast, 16x16 depth tiles, 64-pixel ICP chunks at level 1)."""
    return (200, 152, 99.5, 75.5, 164.0, 164.0)


def odd_cam():
    return (101, 77, 50.0, 38.0, 82.0, 82.0)


def oracle_preprocess(mm, ocam):
    d = O.depth_mm_to_m(mm)
    tr = O.trunc_depth(d, P["depth_trunc_min"], P["depth_trunc_max"])
    fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
    v = O.depth_to_vertices(fl, ocam)
    n = O.vertices_to_normals(v)
    return d, tr, fl, v, n


@pytest.mark.parametrize("cam", [small_cam(), mid_cam(), S.vga_camera(), ragged_cam(), odd_cam()])
def test_preprocess_chain(cam):
    size = 3.0
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    mm = S.render_depth_mm(S.trajectory_pose(7, size), cam, size)
    mm[5:9, 10:20] = 0            # holes
    mm[20, :] = 60000             # beyond trunc_max
    d, tr, fl, v, n = oracle_preprocess(mm, ocam)
    ctx = K.Context(kcam, 32, size, levels=3)
    ctx.upload_depth_mm(mm)
    ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    assert np.array_equal(bits(ctx.download_map(K.MAP_RAW_DEPTH)), bits(d))
    assert np.array_equal(bits(ctx.download_map(K.MAP_TRUNCED_DEPTH)), bits(tr))
    g_fl = ctx.download_map(K.MAP_FILTERED_DEPTH)
    # device __expf vs libm expf: a few ulp on the weights -> 2e-6 relative on the filtered depth
    assert np.allclose(g_fl, fl, rtol=2e-6, atol=0)
    assert np.array_equal(g_fl == 0, fl == 0)
    # vertices / normals are exact functions of the filtered depth: feed the oracle the device's own filtered map
    v2 = O.depth_to_vertices(g_fl, ocam)
    n2 = O.vertices_to_normals(v2)
    assert np.array_equal(bits(ctx.download_map(K.MAP_NEW_VERTICES)), bits(v2))
    assert np.array_equal(bits(ctx.download_map(K.MAP_NEW_NORMALS)), bits(n2))
    # pyramids (new maps), levels 1 and 2
    ctx.downsample(model=False)
    for lvl, (ov, on) in enumerate(zip(O.pyramid(v2, 3), O.pyramid(n2, 3, normals=True))):
        assert np.array_equal(bits(ctx.download_map(K.MAP_NEW_VERTICES, lvl)), bits(ov))
        assert np.array_equal(bits(ctx.download_map(K.MAP_NEW_NORMALS, lvl)), bits(on))
    ctx.close()


def test_empty_and_invalid_depth():
    cam = small_cam()
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ctx = K.Context(kcam, 32, 3.0, levels=3)
    mm = np.zeros((48, 64), np.uint16)
    ctx.upload_depth_mm(mm)
    ctx.preprocess(0.3, 4.0, 2.0, 0.03)
    assert not ctx.download_map(K.MAP_NEW_VERTICES).any() and not ctx.download_map(K.MAP_NEW_NORMALS).any()
    ctx.integrate(S.pose0(3.0), 0.1, 2.0)
    assert ctx.stats()["updated_last"] == 0 and ctx.stats()["weight_gt0"] == 0
    ctx.raycast(S.pose0(3.0), 0.07, 0.3, 4.0)
    assert not ctx.download_map(K.MAP_MODEL_VERTICES).any()
    ctx.close()


def _fuse_sequence(res, size, cam, n_frames, sdf_trunc, max_dist, color=False, levels=3):
    """Integrate n_frames of Scene S with ground-truth poses on both sides; returns (ctx, ovol, last pose, maps)."""
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=levels, max_triangles=400000, has_color=color)
    rng = np.random.default_rng(5)
    pose = None
    for k in range(n_frames):
        pose = S.trajectory_pose(3 * k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        d, tr, fl, v, n = oracle_preprocess(mm, ocam)
        rgb = rng.integers(0, 256, (cam[1], cam[0], 3)).astype(np.uint8) if color else None
        n_o = O.integrate(ovol, tr, n, rgb, color, color, pose, sdf_trunc, max_dist, ocam, ocam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        if color:
            ctx.upload_rgb(rgb)
            ctx.upload_map(K.MAP_NEW_NORMALS, 0, n)      # colour weights read normal.z: use the oracle's normals exactly
        ctx.integrate(pose, sdf_trunc, max_dist, has_color=color, angle_weight=color)
        st = ctx.stats()
        assert st["updated_last"] == n_o, (k, st, n_o)
        assert st["weight_gt0"] == O.count_weight_gt0(ovol)
    return ctx, ovol, pose, ocam


@pytest.mark.parametrize("res,size,cam,trunc", [(32, 3.0, small_cam(), 0.2), (64, 3.0, mid_cam(), 0.1), (128, 3.0, S.vga_camera(), 0.05),
                                                (96, 3.0, ragged_cam(), 0.12), (40, 2.5, odd_cam(), 0.2)])
def test_integrate_raycast_mc_bit_exact(res, size, cam, trunc):
    ctx, ovol, pose, ocam = _fuse_sequence(res, size, cam, 3, trunc, 2.5)
    t, w = ctx.download_volume()
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    inc = 0.7 * trunc
    ov, on, _ = O.raycast(ovol, False, pose, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    ctx.raycast(pose, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    gv, gn = ctx.download_map(K.MAP_MODEL_VERTICES), ctx.download_map(K.MAP_MODEL_NORMALS)
    assert int((ov[..., 3] != 0).sum()) > 100                      # the case exercises real hits
    assert np.array_equal(bits(gv), bits(ov)) and np.array_equal(bits(gn), bits(on))
    # model pyramids from the raycast maps
    ctx.downsample(model=True)
    for lvl, (pv, pn) in enumerate(zip(O.pyramid(ov, 3), O.pyramid(on, 3, normals=True))):
        assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_VERTICES, lvl)), bits(pv))
        assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_NORMALS, lvl)), bits(pn))
    thr = 300 * size / res
    otris = O.marching_cubes(ovol, False, thr, 400000)
    ctx.marching_cubes(thr)
    gtris = ctx.triangles()
    assert len(otris) > 50 and len(gtris) == len(otris)
    assert np.array_equal(gtris["v"]["pos"].view(np.uint32), otris["v"]["pos"].view(np.uint32))   # same sequence, (z,y,x,k) order
    # appending: a second extraction lands after the first (MarchingcubeData never clears its counter)
    ctx.marching_cubes(thr)
    assert len(ctx.triangles()) == 2 * len(otris)
    ctx.clear_triangles()
    assert len(ctx.triangles()) == 0
    ctx.close()


@pytest.mark.parametrize("res,px_per_voxel,z_plane", [(64, 1.0, 1.0), (64, 2.0, 1.5), (96, 0.5, 1.0), (128, 4.0, 2.0)])
def test_integrate_with_voxels_projecting_onto_pixel_boundaries(res, px_per_voxel, z_plane):
    """The cull keeps a brick for the pixels floor(u0 + 0.5 - 1/16) .. floor(u1 + 0.5 + 1/16) its voxels can land on -- no spare pixel on either side
    (cull.h).  Adversarial geometry for that: a camera looking straight down +z whose focal length makes voxel centres of one depth plane project
    EXACTLY onto pixel boundaries (u + 0.5 an integer: the fusion kernel's floor and the cull's bound meet at a tie), a depth image that changes from
    pixel to pixel (so a wrong pixel changes the verdict), several truncation distances and integration distances.  N_upd and every voxel bit must
    equal the oracle's, which visits every voxel."""
    size = 2.0
    cell = size / res
    cols, rows = 96, 80
    f = px_per_voxel * z_plane / cell                      # a voxel step at depth z_plane is px_per_voxel pixels: centres at (i + 0.5) steps -> half-pixel positions
    cam = (cols, rows, cols / 2.0, rows / 2.0, f, f)       # integer principal point: u + 0.5 = (i + 0.5) * px_per_voxel + cx + 0.5
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    rng = np.random.default_rng(res)
    pose = np.eye(4, dtype=np.float32)
    pose[0, 3] = size / 2; pose[1, 3] = size / 2; pose[2, 3] = -(z_plane - size / 2)      # the plane z_plane in front of the camera is the volume's mid plane
    # depth: constant over each 8 x 8 pixel tile (the cull's table granularity), unrelated from tile to tile -- no depth at all, in front of the mid
    # plane, behind it, out of reach -- so a brick whose last pixel column is missed reads the wrong tile and decides wrongly; + pixel noise
    tiles = rng.integers(0, 4, (rows // 8, cols // 8))
    per_tile = np.choose(tiles, [0.0, z_plane - 3.3 * cell, z_plane + 2.7 * cell, z_plane + 40.0 * cell]).astype(np.float32)
    depth = np.kron(per_tile, np.ones((8, 8), np.float32))
    depth = np.where(depth > 0, depth + rng.uniform(-0.4, 0.4, (rows, cols)).astype(np.float32) * cell, 0.0).astype(np.float32)
    normals = np.zeros((rows, cols, 4), np.float32); normals[..., 2] = -1.0
    for trunc, dist in ((2.5 * cell, 10.0), (5.0 * cell, z_plane + 1.1 * cell), (1.01 * cell, 10.0)):
        ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
        ovol = O.OVolume(res, size, P["volume_max_weight"])
        ctx.upload_map(K.MAP_TRUNCED_DEPTH, 0, depth)
        for shift in (0.0, 0.25 * cell, -0.5 * cell):      # the tie itself, and poses a fraction of a voxel off it
            p2 = pose.copy(); p2[0, 3] += shift; p2[1, 3] -= shift
            n_o = O.integrate(ovol, depth, normals, None, False, False, p2, trunc, dist, ocam, ocam)
            ctx.integrate(p2, trunc, dist)
            assert ctx.stats()["updated_last"] == n_o and n_o > 1000, (res, px_per_voxel, trunc, dist, shift, n_o)
        t, w = ctx.download_volume()
        assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
        ctx.close()


def test_triangle_cap():
    ctx, ovol, pose, ocam = _fuse_sequence(32, 3.0, small_cam(), 2, 0.2, 2.5)
    thr = 300 * 3.0 / 32
    full = O.marching_cubes(ovol, False, thr, 400000)
    cap = len(full) // 2
    ctx2 = K.Context(K.camera(*small_cam()), 32, 3.0, P["volume_max_weight"], levels=3, max_triangles=cap)
    t, w = ctx.download_volume()
    ctx2.upload_volume(t, w)
    ctx2.marching_cubes(thr)
    g = ctx2.triangles()
    assert len(g) == cap and np.array_equal(g["v"]["pos"].view(np.uint32), full[:cap]["v"]["pos"].view(np.uint32))
    ctx.close(); ctx2.close()


def test_raycast_after_volume_upload_uses_rebuilt_skip_tables():
    """kf_upload_volume rebuilds the brick flags and the packed macro / super-cell tables the raycast's empty-space walk reads (ctx.hip
    k_rebuild_flags): a volume uploaded into a fresh context, and into a context that held OTHER content before, must raycast to the
    oracle's maps bit for bit; an empty volume uploaded over a used one leaves no stale cell behind (no hit anywhere)."""
    res, size = 128, 3.0
    ctx, ovol, pose, ocam = _fuse_sequence(res, size, small_cam(), 3, 5 * size / res, 2.5)
    inc = 0.8 * 5 * size / res
    ov, on, _ = O.raycast(ovol, False, pose, inc, ocam, 0.3, 4.0)
    assert int((ov[..., 3] != 0).sum()) > 500
    t, w = ctx.download_volume()
    fresh = K.Context(K.camera(*small_cam()), res, size, P["volume_max_weight"], levels=3)
    used = K.Context(K.camera(*small_cam()), res, size, P["volume_max_weight"], levels=3)
    rng = np.random.default_rng(5)
    junk_t = np.where(rng.random(t.shape) < 0.02, -0.5, 1.0).astype(np.float32)          # negative voxels scattered over every macro cell
    used.upload_volume(junk_t, np.ones_like(w))
    used.raycast(pose, inc, 0.3, 4.0)
    for c in (fresh, used):
        c.upload_volume(t, w)
        c.raycast(pose, inc, 0.3, 4.0)
        assert np.array_equal(bits(c.download_map(K.MAP_MODEL_VERTICES)), bits(ov))
        assert np.array_equal(bits(c.download_map(K.MAP_MODEL_NORMALS)), bits(on))
    used.upload_volume(np.zeros_like(t), np.zeros_like(w))
    used.raycast(pose, inc, 0.3, 4.0)
    assert not used.download_map(K.MAP_MODEL_VERTICES).any()
    for c in (ctx, fresh, used):
        c.close()


def test_integrate_color_bit_exact():
    ctx, ovol, pose, ocam = _fuse_sequence(32, 3.0, small_cam(), 3, 0.2, 2.5, color=True)
    t, w, c = ctx.download_volume(color=True)
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    seen = ovol.weight > 0
    assert np.array_equal(c[seen], ovol.color[seen])
    ov, on, orgb = O.raycast(ovol, True, pose, 0.14, ocam, 0.3, 4.0)
    ctx.raycast(pose, 0.14, 0.3, 4.0, has_color=True)
    assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_VERTICES)), bits(ov))
    assert np.array_equal(ctx.download_map(K.MAP_RAYCAST_RGB), orgb)
    thr = 300 * 3.0 / 32
    otris = O.marching_cubes(ovol, True, thr, 400000)
    ctx.marching_cubes(thr, has_color=True)
    g = ctx.triangles()
    assert len(g) == len(otris) and np.array_equal(g.view(np.uint32), otris.view(np.uint32))
    ctx.close()


def _tracking_case(res, size, cam, trunc, n_warm=2, levels=3):
    """Fuse a few frames, raycast from the last pose, then present the NEXT frame: returns everything ICP needs."""
    ctx, ovol, pose, ocam = _fuse_sequence(res, size, cam, n_warm, trunc, 2.5, levels=levels)
    inc = 0.7 * trunc
    ov, on, _ = O.raycast(ovol, False, pose, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    ctx.raycast(pose, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    nxt = S.trajectory_pose(3 * n_warm, size).astype(np.float32)
    mm = S.render_depth_mm(nxt, cam, size)
    d, tr, fl, v, n = oracle_preprocess(mm, ocam)
    ctx.upload_depth_mm(mm)
    ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    # make the inputs of the tracker identical on both sides (bilateral differs in the last ulps)
    ctx.upload_map(K.MAP_NEW_VERTICES, 0, v)
    ctx.upload_map(K.MAP_NEW_NORMALS, 0, n)
    return ctx, ovol, pose, nxt, ocam, (v, n, ov, on, tr)


# truncation = 5 voxels: thinner bands leave the raycast gradient taps at grazing walls unobserved (no model normals there),
# the 6x6 system then loses rank and the fp32 solve of the reference is meaningless on either side
@pytest.mark.parametrize("res,cam,trunc", [(64, mid_cam(), 5 * 3.0 / 64), (128, S.vga_camera(), 5 * 3.0 / 128), (96, ragged_cam(), 5 * 3.0 / 96)])
def test_icp_system_and_track(res, cam, trunc):
    size = 3.0
    ctx, ovol, pose, nxt, ocam, (v, n, ov, on, tr) = _tracking_case(res, size, cam, trunc)
    nv, nn = O.pyramid(v, 3), O.pyramid(n, 3, normals=True)
    mv, mn = O.pyramid(ov, 3), O.pyramid(on, 3, normals=True)
    ctx.downsample(model=False); ctx.downsample(model=True)
    last_inv = O.mat44_inverse(pose)
    ocams = [ocam, ocam.half(), ocam.half().half()]
    kcams = [ctx.cam, K.half_camera(ctx.cam), K.half_camera(K.half_camera(ctx.cam))]
    for lvl in range(3):
        sd, sf, valid = O.icp_system(nv[lvl], nn[lvl], mv[lvl], mn[lvl], ocams[lvl], pose, last_inv, P["icp_thre_dist"], P["icp_thre_sin_angle"])
        g = ctx.icp_system(lvl, pose, last_inv, kcams[lvl], P["icp_thre_dist"], P["icp_thre_sin_angle"])
        assert valid > 50
        # fp32 sums in a different association than the double ground truth: 1e-5 of the largest entry
        assert np.max(np.abs(g - sd)) <= 1e-5 * np.max(np.abs(sd)), (lvl, g, sd)
    ok_o, pose_o = O.icp_estimate(nv, nn, mv, mn, ocam, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    ctx.set_pose(pose)
    ctx.icp_track(1, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    ok_g, pose_g, status, iters = ctx.track_result()
    assert ok_o and ok_g and status == 0 and iters == 19
    # north-star tolerance: 1e-4 m / 1e-4 rad
    assert np.max(np.abs(pose_g[:3, 3] - pose_o[:3, 3])) < 1e-4
    assert np.max(np.abs(pose_g[:3, :3] - pose_o[:3, :3])) < 1e-4
    # and the estimate moved towards the true pose of the new frame
    assert np.linalg.norm(pose_g[:3, 3] - nxt[:3, 3]) < np.linalg.norm(pose[:3, 3] - nxt[:3, 3]) + 1e-3
    # frame 0 never tracks and never changes the pose
    ctx.set_pose(pose)
    ctx.icp_track(0, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    ok0, pose0, _, it0 = ctx.track_result()
    assert ok0 and it0 == 0 and np.array_equal(pose0, pose)
    ctx.close()


@pytest.mark.parametrize("levels,iters", [(1, 3), (2, 15)])
def test_icp_track_with_fewer_pyramid_levels(levels, iters):
    """nPyramidLevels 1 and 2 (src/CameraPoseFinderICP.cpp:14-35: 3 iterations, resp. 10 + 5) against the oracle's loop."""
    size, res, cam, trunc = 3.0, 64, mid_cam(), 5 * 3.0 / 64
    ctx, ovol, pose, nxt, ocam, (v, n, ov, on, tr) = _tracking_case(res, size, cam, trunc, levels=levels)
    nv, nn = O.pyramid(v, levels), O.pyramid(n, levels, normals=True)
    mv, mn = O.pyramid(ov, levels), O.pyramid(on, levels, normals=True)
    ok_o, pose_o = O.icp_estimate(nv, nn, mv, mn, ocam, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    ctx.set_pose(pose)
    ctx.icp_track(1, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    ok_g, pose_g, status, it = ctx.track_result()
    assert ok_o and ok_g and status == 0 and it == iters
    assert np.max(np.abs(pose_g[:3, 3] - pose_o[:3, 3])) < 1e-4 and np.max(np.abs(pose_g[:3, :3] - pose_o[:3, :3])) < 1e-4
    ctx.close()


def test_icp_lost_keeps_pose():
    """Shake threshold 0 -> the first step is rejected: findCameraPose false, pose unchanged, integrate skipped."""
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(64, 3.0, mid_cam(), 5 * 3.0 / 64)
    ctx.set_pose(pose)
    before = ctx.stats()["weight_gt0"]
    ctx.icp_track(1, P["icp_thre_dist"], P["icp_thre_sin_angle"], 0.0, 0.0)
    ok, pose_g, status, iters = ctx.track_result()
    assert not ok and status == 2 and iters == 0 and np.array_equal(pose_g, pose)
    ctx.integrate(None, 0.1, 2.5)                   # device-resident pose path: skipped because tracking failed
    st = ctx.stats()
    assert st["updated_last"] == 0 and st["weight_gt0"] == before
    ctx.close()


@pytest.mark.parametrize("cam", [mid_cam(), S.vga_camera()])
def test_icp_on_a_single_plane_is_lost_by_the_determinant_test_on_both_sides(cam):
    """A frame that sees nothing but one fronto-parallel plane constrains three of the six degrees of freedom: every row of the system is
    (-y, x, 0, 0, 0, -1) . (rotation, translation), so three rows and columns of J^T J are EXACTLY zero.  The reference gives the frame up at its
    determinant test (ICP.cpp:138, determinant 0 < 1e-10); here the determinant is the product of the Cholesky pivots and a pivot <= 0 counts as
    singular (llt_solve6_wave): the same verdict, status 1, no Gauss-Newton step applied, pose untouched -- in the persistent loop and, with a
    second context alive, in the per-step form."""
    cols, rows = cam[0], cam[1]
    ocam = O.Cam.make(*cam)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    z = np.float32(1.5)
    v = np.zeros((rows, cols, 4), np.float32)
    v[..., 0] = (xx - cam[2]) / cam[4] * z; v[..., 1] = (yy - cam[3]) / cam[5] * z; v[..., 2] = z; v[..., 3] = 1.0
    n = np.zeros((rows, cols, 4), np.float32); n[..., 2] = -1.0
    pose = np.eye(4, dtype=np.float32)
    nv, nn = O.pyramid(v, 3), O.pyramid(n, 3, normals=True)
    ok_o, pose_o = O.icp_estimate(nv, nn, nv, nn, ocam, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    assert not ok_o and np.array_equal(pose_o, pose)
    for second_context in (False, True):
        other = K.Context(K.camera(*small_cam()), 32, 3.0, levels=3) if second_context else None
        ctx = K.Context(K.camera(*cam), 64, 3.0, levels=3)
        ctx.set_pose(pose)
        for m_v, m_n in ((K.MAP_NEW_VERTICES, K.MAP_NEW_NORMALS), (K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)):
            ctx.upload_map(m_v, 0, v); ctx.upload_map(m_n, 0, n)
        ctx.icp_track(1, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        ok_g, pose_g, status, iters = ctx.track_result()
        assert not ok_g and status == 1 and iters == 0 and np.array_equal(pose_g, pose), (second_context, ok_g, status, iters)
        assert ctx.last_form == (2 if second_context else 1) or not default_forms()
        ctx.close()
        if other is not None:
            other.close()


def test_sdf_system_and_track():
    size, res, cam, trunc = 3.0, 64, mid_cam(), 5 * 3.0 / 64
    ctx, ovol, pose, nxt, ocam, (v, n, ov, on, tr) = _tracking_case(res, size, cam, trunc, n_warm=3)
    sd, sf, valid = O.sdf_system(ovol, tr, ocam, pose)
    g = ctx.sdf_system(pose)
    assert valid > 200
    assert np.max(np.abs(g - sd)) <= 1e-5 * np.max(np.abs(sd)), (g, sd)
    ok_o, pose_o, it_o = O.sdf_estimate(ovol, tr, ocam, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    ctx.set_pose(pose)
    ctx.sdf_track(1, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])
    ok_g, pose_g, status, iters = ctx.track_result()
    assert ok_o == ok_g
    if ok_o:
        assert iters == it_o
        assert np.max(np.abs(pose_g - pose_o)) < 1e-4
    ctx.close()


def test_full_size_integrate_counts_512():
    """BASELINE config C2 geometry (512^3 @ 4 m, VGA): update count and observed-voxel count equal the oracle's;
    linearity property: integrating the same frame twice doubles no count but saturates weights identically."""
    size, res, cam = 4.0, 512, S.vga_camera()
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    for k in (0, 5):
        pose = S.trajectory_pose(k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        d, tr, fl, v, n = oracle_preprocess(mm, ocam)
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"], ocam, ocam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ctx.integrate(pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"])
        st = ctx.stats()
        assert st["updated_last"] == n_o and n_o > 1000000
        assert st["weight_gt0"] == O.count_weight_gt0(ovol)
        assert st["bricks_active"] < st["bricks_total"] // 4          # the cull pass really skips most of the volume
    # checksum of the planes over the slab the camera sees
    t, w = ctx.download_volume(0, 256)
    assert np.array_equal(bits(t), bits(ovol.tsdf[:256])) and np.array_equal(bits(w), bits(ovol.weight[:256]))
    ctx.close()


def test_exact_division_helper():
    """The kernels divide through a split exact-division helper (shared reciprocal); it must agree bit for bit with the
    compiler's IEEE `/` on the operand ranges the kernels use -- 4 x 2^24 random quotients."""
    ctx = K.Context(K.camera(*small_cam()), 32, 3.0, levels=3)
    for mode in range(4):
        assert ctx.selftest_div(1 << 24, 12345 + mode, mode) == 0, mode
    # the float / double -> int conversion as ONE hardware instruction (v_cvt_i32_f32 / _f64: truncating, saturating, NaN -> 0) against its
    # spelled-out definition -- the reference's CUDA conversion rule (cvt.rzi.s32): special values + 2^24 random bit patterns each
    for mode in (10, 11):
        assert ctx.selftest_div(1 << 24, 777 + mode, mode) == 0, mode
    # the ray parameter's walk in closed form (kf_ray_advance) against the chain of fp32 additions it stands for (the reference's
    # `ray_current += fRayIncrement`, raycastingVolume.cu:116): 2^22 random (start, increment, exit) triples incl. binade crossings and ties
    for seed in (1, 2):
        assert ctx.selftest_div(1 << 22, 4242 + seed, 12) == 0, seed
    ctx.close()


def test_prefetched_preprocess_is_bit_identical():
    """kf_prefetch_frame (next frame preprocessed on a side stream while the current one is tracked) must change nothing:
    same maps, same poses, same volume as the plain call sequence -- including when the prefetched frame is NOT the one that
    comes next (the prefetch is then ignored)."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    res, size, cam = 384, 3.0, S.vga_camera()                    # stock truncation (0.05 m) = 6.4 voxels, as at C2
    wl = dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"])
    n = 6
    frames = np.stack([S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    outs = []
    forms = {}
    for mode in ("plain", "prefetch", "wrong-prefetch", "prefetch-per-step"):
        # the request is honoured by the raycast launch whichever form the tracker took before it: the persistent loop, or -- here forced by a
        # second live context -- one launch per Gauss-Newton step
        other_ctx = K.Context(K.camera(*small_cam()), 32, 3.0, levels=3) if mode == "prefetch-per-step" else None
        pipe = SingleGpuPipeline(K.camera(*cam), res, size, wl)
        poses = []
        for k in range(n):
            nxt = None
            if mode.startswith("prefetch") and k + 1 < n:
                nxt = dev.data_ptr() + (k + 1) * fb
            if mode == "wrong-prefetch":
                nxt = dev.data_ptr() + ((k + 3) % n) * fb
            pipe.process_frame_device(dev.data_ptr() + k * fb, k, nxt)
            ok, pose, status, iters = pipe.track_result()
            assert ok
            poses.append(pose.copy())
            forms.setdefault(mode, set()).add(pipe.ctx.last_form)
        pipe.sync()
        if other_ctx is not None:
            other_ctx.close()
        maps = [pipe.ctx.download_map(m) for m in (K.MAP_RAW_DEPTH, K.MAP_TRUNCED_DEPTH, K.MAP_FILTERED_DEPTH, K.MAP_NEW_VERTICES,
                                                   K.MAP_NEW_NORMALS, K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)]
        vol = pipe.ctx.download_volume()
        outs.append((poses, maps, vol))
        pipe.close()
    assert (1 in forms["prefetch"] and 2 in forms["prefetch-per-step"] and 1 not in forms["prefetch-per-step"]) or not default_forms()     # persistent loop / one launch per step
    for other in outs[1:]:
        for a, b in zip(outs[0][0], other[0]):
            assert np.array_equal(a, b)
        for a, b in zip(outs[0][1], other[1]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(outs[0][2][0], other[2][0]) and np.array_equal(outs[0][2][1], other[2][1])


def test_host_stream_staged_two_ahead_equals_the_resident_stream():
    """kf_upload_depth_mm_next / kf_take_next_depth (frames staged from HOST memory ahead of their use, their front ends riding in the previous
    frame's launches -- SingleGpuPipeline.process_frame_host) change nothing: poses, maps and voxels equal the device-resident stream's bit for
    bit, to the end of the stream (the last frames have nothing staged behind them), and after a restart in the middle (a frame id that does
    not follow).  The staging states answer as include/hybkf.h says: a third staged frame and a take with nothing staged are KF_ERR_STATE."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    res, size, cam = 256, 3.0, S.vga_camera()
    wl = dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"])
    n = 9
    frames = np.stack([S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    outs = []
    for mode in ("resident", "host", "host-restart"):
        pipe = SingleGpuPipeline(K.camera(*cam), res, size, wl)
        poses = []
        for k in range(n):
            if mode == "resident":
                pipe.process_frame_device(dev.data_ptr() + k * fb, k, None)
            else:
                if mode == "host-restart" and k == 5:
                    pipe._host = None                                     # what a seek in the stream does
                pipe.process_frame_host(lambda j: frames[j] if j < n else None, k)
            ok, pose, status, iters = pipe.track_result()
            assert ok
            poses.append(pose.copy())
        pipe.sync()
        maps = [pipe.ctx.download_map(m) for m in (K.MAP_RAW_DEPTH, K.MAP_FILTERED_DEPTH, K.MAP_NEW_VERTICES, K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)]
        outs.append((poses, maps, pipe.ctx.download_volume()))
        if mode == "host":
            c = pipe.ctx
            with pytest.raises(K.KfError):
                c.take_next_depth()                                       # the stream has ended: nothing is staged
            c.upload_depth_mm_next(frames[0]); c.upload_depth_mm_next(frames[1])
            with pytest.raises(K.KfError):
                c.upload_depth_mm_next(frames[2])                         # two frames ahead is the limit
            c.upload_depth_mm(frames[3])                                  # drops both
            with pytest.raises(K.KfError):
                c.take_next_depth()
        pipe.close()
    for other in outs[1:]:
        for a, b in zip(outs[0][0], other[0]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for a, b in zip(outs[0][1], other[1]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(outs[0][2][0].view(np.uint32), other[2][0].view(np.uint32)) and np.array_equal(outs[0][2][1], other[2][1])


def test_lost_frame_in_the_prefetched_stream_equals_the_plain_sequence():
    """The steady-state frame has no pyramid / set-up launch of its own (the pyramids ride in the previous raycast launch, the persistent loop
    writes its whole verdict itself): a frame the tracker LOSES in the middle of a prefetched stream, and the frames after it, must leave the
    same verdicts, pose bits, counters and voxel bits as the plain call sequence, which runs the set-up launch before every loop."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    res, size, cam = 256, 3.0, S.vga_camera()
    wl = dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"])
    n, bad = 8, 4
    frames = np.stack([S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)])
    frames[bad] = 0                                                        # a sensor drop-out: no valid pixel, no system to solve (ICP.cpp:138)
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    outs = []
    for mode in ("plain", "prefetch"):
        pipe = SingleGpuPipeline(K.camera(*cam), res, size, wl)
        log = []
        for k in range(n):
            nxt = dev.data_ptr() + (k + 1) * fb if (mode == "prefetch" and k + 1 < n) else None
            pipe.process_frame_device(dev.data_ptr() + k * fb, k, nxt)
            ok, pose, status, iters = pipe.track_result()
            log.append((bool(ok), int(status), int(iters), pose.copy()))
        pipe.sync()
        st = pipe.stats()
        outs.append((log, pipe.ctx.download_volume(), st["frames_fused"], st["frames_lost"], pipe.ctx.download_map(K.MAP_MODEL_VERTICES)))
        pipe.close()
    (la, va, fa, xa, ma), (lb, vb, fb_, xb, mb) = outs
    assert not la[bad][0] and la[bad][1] != 0 and all(r[0] for i, r in enumerate(la) if i != bad)       # exactly the empty frame is lost
    assert fa == n - 1 and xa == 1 and (fa, xa) == (fb_, xb)
    for ra, rb in zip(la, lb):
        assert ra[:3] == rb[:3] and np.array_equal(ra[3].view(np.uint32), rb[3].view(np.uint32))
    assert np.array_equal(va[0].view(np.uint32), vb[0].view(np.uint32)) and np.array_equal(va[1], vb[1])
    assert np.array_equal(ma.view(np.uint32), mb.view(np.uint32))


def test_maximum_configuration_2048_cubed():
    """BASELINE.json's largest configuration on ONE GPU: 2048^3 @ 8 m (68.7 GB of voxels), 1280x960 depth.  Exercises the 64-bit
    voxel addressing, a brick queue of > 1 M entries, the one-launch-per-step ICP (1.2 M pixels exceed the persistent loop's
    workgroup budget) and the 2-D marching-cubes grid (33.5 M blocks)."""
    import ctypes as C
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2**30:
        pytest.skip("needs ~75 GB of free HBM")
    res, size, cam = 2048, 8.0, S.vga_camera(2)
    n = 4
    frames = np.stack([S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SingleGpuPipeline(K.camera(*cam), res, size, dict(trunc_max=8.0, integ_dist=8.0), max_triangles=16_000_000)
    for k in range(n):
        pipe.process_frame_device(dev.data_ptr() + k * fb, k)
        ok, pose, status, iters = pipe.track_result()
        assert ok and status == 0 and iters == (0 if k == 0 else 19)
        assert np.linalg.norm(pose[:3, 3] - S.trajectory_pose(k, size)[:3, 3]) < 2e-3
    st = pipe.stats(observed=True)
    assert st["frames_fused"] == n and st["frames_lost"] == 0
    assert st["updated_last"] > 4e8 and st["weight_gt0"] >= st["updated_last"] and st["bricks_active"] > 100_000      # (whole free-space bricks are retired by the cull from the second frame on: counted, not queued)
    hit = pipe.ctx.download_map(K.MAP_MODEL_VERTICES)[..., 3] != 0
    assert hit.sum() > 0.9 * hit.size
    # the raycast's two gradient forms at this size (csrc/grad_shared.h: the gathers' view is 16 brick layers deep here, so waves on depth discontinuities take
    # the fallback): identical maps from the same volume and pose; KF_RAYCAST_SHARED_GRAD is read per launch for exactly this
    import os
    maps = {}
    for mode in ("1", "0", "2"):
        os.environ["KF_RAYCAST_SHARED_GRAD"] = mode
        try:
            pipe.ctx.raycast(pose, 0.035, P["depth_trunc_min"], 8.0)
        finally:
            del os.environ["KF_RAYCAST_SHARED_GRAD"]
        maps[mode] = (pipe.ctx.download_map(K.MAP_MODEL_VERTICES), pipe.ctx.download_map(K.MAP_MODEL_NORMALS))
    assert (np.abs(maps["1"][1][..., :3]).sum(axis=-1) > 0).sum() > 0.8 * hit.size
    for mode in ("0", "2"):
        assert np.array_equal(bits(maps["1"][0]), bits(maps[mode][0])) and np.array_equal(bits(maps["1"][1]), bits(maps[mode][1])), mode
    pipe.ctx.marching_cubes(300 * size / res)
    cnt = C.c_uint32()
    assert pipe.ctx.lib.kf_triangle_count(pipe.ctx.h, C.byref(cnt)) == 0
    assert 1_000_000 < cnt.value < 16_000_000
    pipe.close()


def test_weight_saturation_bit_exact():
    """tsdfVolume.h:65 `fminf(weight + 1, max_weight)`: with max_weight = 2 the running average stops adapting after two frames;
    five integrations must still match the oracle bit for bit (tsdf, weight, update counts)."""
    size, res, cam, trunc = 3.0, 64, mid_cam(), 0.1
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ovol = O.OVolume(res, size, 2.0)
    ctx = K.Context(kcam, res, size, 2.0, levels=3)
    for k in range(5):
        pose = S.trajectory_pose(4 * k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        d, tr, fl, v, n = oracle_preprocess(mm, ocam)
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ctx.integrate(pose, trunc, 2.5)
        assert ctx.stats()["updated_last"] == n_o
    t, w = ctx.download_volume()
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    assert w.max() == 2.0 and (w == 2.0).sum() > 1000
    ctx.close()


def test_reset_volume_equals_fresh_context():
    """kf_reset_volume: voxels, brick flags, skip tables and counters all return to their initial state -- a sequence fused after a
    reset is bit-identical (volume, raycast maps, triangles) to the same sequence on a new context."""
    size, res, cam, trunc = 3.0, 64, mid_cam(), 0.1
    kcam = K.camera(*cam)

    def fuse(ctx, ks):
        for k in ks:
            pose = S.trajectory_pose(3 * k, size).astype(np.float32)
            ctx.upload_depth_mm(S.render_depth_mm(pose, cam, size))
            ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
            ctx.integrate(pose, trunc, 2.5)
        ctx.raycast(pose, 0.7 * trunc, P["depth_trunc_min"], P["depth_trunc_max"])
        ctx.marching_cubes(300 * size / res)
        return ctx.download_volume(), ctx.download_map(K.MAP_MODEL_VERTICES), ctx.download_map(K.MAP_MODEL_NORMALS), ctx.triangles()

    a = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=200000)
    fuse(a, [5, 6, 7])                                            # something else first
    a.reset_volume()
    st = a.stats()
    assert st["weight_gt0"] == 0 and st["updated_total"] == 0 and st["frames_fused"] == 0
    t, w = a.download_volume()
    assert not t.any() and not w.any()
    a.raycast(S.trajectory_pose(0, size).astype(np.float32), 0.7 * trunc, P["depth_trunc_min"], P["depth_trunc_max"])
    assert not a.download_map(K.MAP_MODEL_VERTICES).any()
    a.clear_triangles()
    got = fuse(a, [0, 1, 2])
    b = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=200000)
    want = fuse(b, [0, 1, 2])
    assert np.array_equal(bits(got[0][0]), bits(want[0][0])) and np.array_equal(bits(got[0][1]), bits(want[0][1]))
    assert np.array_equal(bits(got[1]), bits(want[1])) and np.array_equal(bits(got[2]), bits(want[2]))
    assert len(got[3]) == len(want[3]) > 1000 and got[3].tobytes() == want[3].tobytes()
    a.close(); b.close()


def test_tile_tables_follow_every_call_order():
    """The integrate cull reads per-tile depth maxima that normally ride in the fused preprocess kernel (built for the distance the
    previous integrate used) and are cleared by the fusion pass.  Whatever the call order -- a changed distance, two integrates of
    one frame, two preprocesses before an integrate, the per-call wrappers, an uploaded depth map -- the update count must equal
    the oracle's (the fallback rebuilds the tables whenever they do not describe the current depth map and distance)."""
    size, res, cam = 3.0, 64, mid_cam()
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    trunc = 0.1

    def frame(k):
        pose = S.trajectory_pose(2 * k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        tr, fl, v, n = oracle_preprocess(mm, ocam)[1:]
        return pose, mm, tr, n

    def both(pose, tr, n, dist):
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, trunc, dist, ocam, ocam)
        ctx.integrate(pose, trunc, dist)
        assert ctx.stats()["updated_last"] == n_o and n_o > 1000, (dist, n_o)

    def pre(mm):
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])

    pose, mm, tr, n = frame(0); pre(mm); both(pose, tr, n, 2.5)            # first call: no distance known yet -> fallback
    pose, mm, tr, n = frame(1); pre(mm); both(pose, tr, n, 2.5)            # tables built by the preprocess kernel
    both(pose, tr, n, 2.5)                                                  # same frame again: tables were cleared -> fallback
    pose, mm, tr, n = frame(2); pre(mm); both(pose, tr, n, 1.5)            # built for 2.5, asked for 1.5 -> fallback
    pose, mm, tr, n = frame(3); pre(mm); both(pose, tr, n, 1.5)            # built for 1.5
    pose, mm, tr, n = frame(4); pre(mm); pre(mm); both(pose, tr, n, 1.5)   # second preprocess finds the tables in use
    pose, mm, tr, n = frame(5)
    ctx.upload_depth_mm(mm); ctx.trunc_depth(P["depth_trunc_min"], P["depth_trunc_max"])   # per-call wrapper: no tables
    both(pose, tr, n, 1.5)
    pose, mm, tr, n = frame(6); pre(mm)
    pose7, mm7, tr7, n7 = frame(7)
    ctx.upload_map(K.MAP_TRUNCED_DEPTH, 0, tr7)                             # the depth map changes behind the tables' back
    both(pose7, tr7, n7, 1.5)
    t, w = ctx.download_volume()
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    st = ctx.stats()
    assert st["frames_fused"] == 8 and st["weight_gt0"] == O.count_weight_gt0(ovol)
    ctx.close()


def test_work_counters_match_the_oracle_march():
    """kf_read_work_counters (the roofline accounting of SURVEY section 8d): the raycast counter equals the number of samples the
    ORACLE's march takes (its per-pixel step counts), the hit counter the number of rays whose crossing was evaluated, and the
    marching-cubes counters the bricks read / triangles produced."""
    res, size, cam, trunc = 128, 3.0, S.vga_camera(), 0.05          # 640x480 is a multiple of the raycast's 32x16 tile
    ctx, ovol, pose, ocam = _fuse_sequence(res, size, cam, 3, trunc, 2.5)
    inc = 0.7 * trunc
    ov, on, _, steps = O.raycast(ovol, False, pose, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"], want_steps=True)
    ctx.stage_timers((1 << 7) | (1 << 6) | (1 << 16))
    ctx.raycast(pose, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    ctx.clear_triangles()
    ctx.marching_cubes(300 * size / res)
    n_steps, n_hits, n_bricks, n_tris = ctx.work_counters()
    total = int(np.asarray(steps, np.int64).sum())
    assert total > 1_000_000
    assert abs(n_steps - total) <= 0.001 * total + cam[0] * cam[1]      # the device derives the count from t: +-1 per ray at most
    assert n_tris == len(O.marching_cubes(ovol, False, 300 * size / res, 400000)) > 1000
    assert 0 < n_bricks <= (res // 8) ** 3
    assert n_hits >= int((ov[..., 3] != 0).sum()) > 10000              # hits = crossings evaluated (some of them fail their taps)
    ms, cnt = ctx.read_stage_ms()
    assert cnt[7] == 1 and cnt[6] == 1 and ms[7] > 0 and ms[6] > 0
    ctx.stage_timers(0)
    ctx.close()


@pytest.mark.parametrize("angled", [False, True])
def test_integrate_color_vga_color_camera_bit_exact(angled):
    """The colour path where it actually lands: the reference projects voxels into the colour image with its literal 525 / 320 / 240
    intrinsics (integrateVolume.cu:56-57), so only a VGA-sized colour camera sees most of them (test_integrate_color_bit_exact's 64x48
    colour image rejects almost every voxel at the window test).  Packed-pair colour fusion (k_integrate_pairs<.., COLOR>) against the oracle:
    update counts, tsdf / weight planes, colour bytes of every observed voxel, raycast colour map; with and without the angle weight; three
    frames so that the running average sees old weights 0, 1, 2."""
    size, res, cam = 3.0, 64, mid_cam()
    rcam = (640, 480, 319.5, 239.5, 525.0, 525.0)
    ocam, kcam, orcam, krcam = O.Cam.make(*cam), K.camera(*cam), O.Cam.make(*rcam), K.camera(*rcam)
    trunc = 5 * size / res
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, has_color=True, rgb_cam=krcam)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    rng = np.random.default_rng(11)
    n_colored = 0
    for k in range(3):
        pose = S.trajectory_pose(3 * k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        d, tr, fl, v, n = oracle_preprocess(mm, ocam)
        rgb = rng.integers(0, 256, (480, 640, 3)).astype(np.uint8)
        n_o = O.integrate(ovol, tr, n, rgb, True, angled, pose, trunc, 2.5, ocam, orcam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ctx.upload_map(K.MAP_NEW_NORMALS, 0, n)                 # identical normals for the angle weight (the bilateral differs in last bits)
        ctx.upload_rgb(rgb)
        ctx.integrate(pose, trunc, 2.5, has_color=True, angle_weight=angled)
        assert ctx.stats()["updated_last"] == n_o and n_o > 10000
        n_colored += n_o
    t, w, c = ctx.download_volume(color=True)
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    seen = ovol.weight > 0
    assert np.array_equal(c[seen], ovol.color[seen]) and int(np.count_nonzero(ovol.color[seen])) > 10000
    ov, on, orgb = O.raycast(ovol, True, pose, 0.7 * trunc, ocam, 0.3, 4.0)
    ctx.raycast(pose, 0.7 * trunc, 0.3, 4.0, has_color=True)
    assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_VERTICES)), bits(ov))
    assert np.array_equal(ctx.download_map(K.MAP_RAYCAST_RGB), orgb) and int(np.count_nonzero(orgb)) > 1000
    ctx.close()


def test_noisy_scene_parity():
    """The frames of bench.py's `scene_noise` block (Scene S + scene.add_sensor_noise: LCG seed 12345, depth-dependent axial noise, 2 % drop-outs)
    at C2's image size: preprocess chain (bilateral early returns, DataPreprocesser.cu:66-69, now fire all over the image), three fused frames
    (update counts and planes bit-exact: partial waves around every drop-out), the raycast maps bit-exact, and one ICP step sequence on identical
    maps within the north star's 1e-4."""
    cam, size, res, trunc = S.vga_camera(), 4.0, 256, 5 * 4.0 / 256
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    clean, poses = S.make_stream(4, cam, size)
    noisy = S.add_sensor_noise(clean)
    assert 0.015 < float((noisy == 0).mean()) < 0.03 and not np.array_equal(noisy[0], clean[0])
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    early = 0
    for k in range(3):
        pose = poses[k].astype(np.float32)
        d, tr, fl, v, n = oracle_preprocess(noisy[k], ocam)
        early += int(((fl == tr) & (tr != 0)).sum())                    # pixels the filter returned unfiltered (or could not change)
        ctx.upload_depth_mm(noisy[k])
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        assert np.array_equal(bits(ctx.download_map(K.MAP_TRUNCED_DEPTH)), bits(tr))
        g_fl = ctx.download_map(K.MAP_FILTERED_DEPTH)
        assert np.allclose(g_fl, fl, rtol=2e-6, atol=0) and np.array_equal(g_fl == 0, fl == 0)
        assert abs(int(((g_fl == tr) & (tr != 0)).sum()) - int(((fl == tr) & (tr != 0)).sum())) <= 8     # the early return fires on the same pixels (a filtered value may coincide with its input in the last bit on one side only)
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam)
        ctx.integrate(pose, trunc, 2.5)
        assert ctx.stats()["updated_last"] == n_o and n_o > 500_000
    assert early > 1000
    t, w = ctx.download_volume()
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    inc = 0.7 * trunc
    ov, on, _ = O.raycast(ovol, False, pose, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    ctx.raycast(pose, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    assert int((ov[..., 3] != 0).sum()) > 50_000
    assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_VERTICES)), bits(ov)) and np.array_equal(bits(ctx.download_map(K.MAP_MODEL_NORMALS)), bits(on))
    # the next noisy frame against that model: identical tracker inputs on both sides
    d, tr, fl, v, n = oracle_preprocess(noisy[3], ocam)
    ctx.upload_depth_mm(noisy[3])
    ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.upload_map(K.MAP_NEW_VERTICES, 0, v)
    ctx.upload_map(K.MAP_NEW_NORMALS, 0, n)
    ok_o, pose_o = O.icp_estimate(O.pyramid(v, 3), O.pyramid(n, 3, normals=True), O.pyramid(ov, 3), O.pyramid(on, 3, normals=True), ocam,
                                  P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    ctx.set_pose(pose)
    ctx.icp_track(1, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    ok_g, pose_g, status, iters = ctx.track_result()
    assert ok_o and ok_g and status == 0 and iters == 19
    assert np.max(np.abs(pose_g[:3, 3] - pose_o[:3, 3])) < 1e-4 and np.max(np.abs(pose_g[:3, :3] - pose_o[:3, :3])) < 1e-4
    assert np.linalg.norm(pose_g[:3, 3] - poses[3][:3, 3]) < 0.01
    ctx.close()


def test_observed_voxel_count_is_kept_running_and_rebased():
    """kf_get_volume_stats.weight_gt0 (the reference prints the count per frame, integrateVolume.cu:91-94): asked every frame, the fusion launches switch to
    their COUNT instantiations and the call becomes a read-back; an upload, a reset, a colour-less scalar launch or 64 frames without a question send it back
    to a sweep.  Every answer equals the sweep (kf_count_observed_voxels) and the oracle's count."""
    size, res, cam = 3.0, 96, mid_cam()
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    trunc = 5 * size / res
    for k in range(12):
        pose = S.trajectory_pose(3 * k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        d, tr, fl, v, n = oracle_preprocess(mm, ocam)
        O.integrate(ovol, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ctx.integrate(pose, trunc, 2.5)
        if k in (5, 6):
            continue                                            # two frames nobody asks about: the running count must still be right afterwards
        st = ctx.stats()                                        # (conftest: every stats() call is cross-checked against the sweep)
        assert st["weight_gt0"] == O.count_weight_gt0(ovol) == ctx.count_observed_voxels(), k
        if k == 8:                                              # an upload writes weights behind the count's back: re-based by the next question
            t, w = ctx.download_volume()
            w[:4] = 0.0; t[:4] = 0.0
            ctx.upload_volume(t, w)
            ovol.tsdf[:4] = 0.0; ovol.weight[:4] = 0.0
            assert ctx.stats()["weight_gt0"] == O.count_weight_gt0(ovol)
    ctx.reset_volume()
    assert ctx.stats()["weight_gt0"] == 0
    ctx.close()
