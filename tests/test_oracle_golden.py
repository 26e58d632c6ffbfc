"""CPU-only: the oracle against the committed golden fixtures and against the reference run recorded in SURVEY.md."""
import os

import numpy as np
import pytest

import oracle_lib as O
from hybkinectfu_amd import scene as S

P = S.STOCK
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_reference_recorded_smoke_numbers():
    """SURVEY.md section 8c records what the reference's own kernels produced (host-compiled in the survey session) for a
    plane at 1.5 m, 256^3 @ 3 m, VGA, pose0: 853 340 voxels with weight>0, 287 680 raycast pixels, centre vertex
    (1.5014, 1.5014, 1.2000), normal (0,0,-1), 34 656 triangles.  The restatement must reproduce them exactly."""
    cam = S.vga_camera()
    ocam = O.Cam.make(*cam)
    size, res = 3.0, 256
    mm = S.render_depth_mm(None, cam, size, plane_depth=1.5)
    tr = O.trunc_depth(O.depth_mm_to_m(mm), 0.3, 4.0)
    fl = O.bilateral(tr, 2.0, 0.03)
    n = O.vertices_to_normals(O.depth_to_vertices(fl, ocam))
    vol = O.OVolume(res, size, 128.0)
    pose = S.pose0(size)
    n_upd = O.integrate(vol, tr, n, None, False, False, pose, 0.05, 2.0, ocam, ocam)
    assert n_upd == 853340 and O.count_weight_gt0(vol) == 853340
    mv, mn, _ = O.raycast(vol, False, pose, 0.7 * 0.05, ocam, 0.3, 4.0)
    assert int((mv[..., 3] != 0).sum()) == 287680
    assert np.allclose(mv[240, 320, :3], [1.5014, 1.5014, 1.2000], atol=5e-5)
    assert np.allclose(mn[240, 320, :3], [0, 0, -1], atol=1e-6)
    assert len(O.marching_cubes(vol, False, 300 * size / res, 6500000)) == 34656


@pytest.mark.parametrize("name,res,cam,trunc", [("s32", 32, (64, 48, 31.5, 23.5, 52.5, 52.5), 5 * 3.0 / 32),
                                                ("s64", 64, (160, 120, 79.5, 59.5, 131.25, 131.25), 5 * 3.0 / 64)])
def test_oracle_matches_golden(name, res, cam, trunc):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    size = 3.0
    ocam = O.Cam.make(*cam)
    vol = O.OVolume(res, size, P["volume_max_weight"])
    n_frames = len(g["n_upd"])
    for k in range(n_frames):
        pose, mm = g["pose%d" % k], g["mm%d" % k]
        assert np.array_equal(mm, S.render_depth_mm(S.trajectory_pose(3 * k, size), cam, size))      # scene generator is frozen too
        tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
        fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        v = O.depth_to_vertices(fl, ocam)
        n = O.vertices_to_normals(v)
        assert O.integrate(vol, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam) == g["n_upd"][k]
    assert np.array_equal(bits(fl), bits(g["filtered_last"]))
    assert np.array_equal(bits(v), bits(g["vertices_last"])) and np.array_equal(bits(n), bits(g["normals_last"]))
    assert np.array_equal(bits(O.pyramid(v, 3)[2]), bits(g["v_l2"]))
    assert np.array_equal(bits(O.pyramid(n, 3, normals=True)[2]), bits(g["n_l2"]))
    assert np.array_equal(bits(vol.tsdf), bits(g["tsdf"])) and np.array_equal(bits(vol.weight), bits(g["weight"]))
    mv, mn, _ = O.raycast(vol, False, pose, 0.7 * trunc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    assert np.array_equal(bits(mv), bits(g["model_v"])) and np.array_equal(bits(mn), bits(g["model_n"]))
    tris = O.marching_cubes(vol, False, 300 * size / res, 400000)
    assert np.array_equal(tris["v"]["pos"].view(np.uint32), g["tri_pos"].view(np.uint32))
    mm = g["mm_next"]
    tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
    fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
    v = O.depth_to_vertices(fl, ocam)
    n = O.vertices_to_normals(v)
    sd, sf, valid = O.icp_system(v, n, mv, mn, ocam, pose, O.mat44_inverse(pose), P["icp_thre_dist"], P["icp_thre_sin_angle"])
    assert valid == g["icp_valid"][0] and np.array_equal(sd, g["icp27"])
    assert np.max(np.abs(sf - sd)) <= 2e-5 * np.max(np.abs(sd))
    ok, p1 = O.icp_estimate(O.pyramid(v, 3), O.pyramid(n, 3, True), O.pyramid(mv, 3), O.pyramid(mn, 3, True), ocam,
                            P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
    assert int(ok) == g["icp_ok"][0] and np.allclose(p1, g["icp_pose"], atol=1e-6)
    sd2, _, valid2 = O.sdf_system(vol, tr, ocam, pose)
    assert valid2 == g["sdf_valid"][0] and np.array_equal(sd2, g["sdf27"])


def test_round_trip_properties():
    """Domain properties that hold at any size: integrate -> raycast recovers the rendered surface; a second integration
    of the same frame leaves tsdf unchanged where weight grew (running average of equal samples); slab-wise integration
    and slab-wise marching cubes concatenate to the whole-volume result."""
    cam = (160, 120, 79.5, 59.5, 131.25, 131.25)
    ocam = O.Cam.make(*cam)
    size, res, trunc = 3.0, 64, 0.1
    pose = S.trajectory_pose(0, size).astype(np.float32)
    mm = S.render_depth_mm(pose, cam, size)
    tr = O.trunc_depth(O.depth_mm_to_m(mm), 0.3, 4.0)
    n = O.vertices_to_normals(O.depth_to_vertices(tr, ocam))
    whole = O.OVolume(res, size, 128.0)
    O.integrate(whole, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam)
    slabs = O.OVolume(res, size, 128.0)
    total = sum(O.integrate(slabs, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam, z0, z0 + 16) for z0 in range(0, 64, 16))
    assert total == O.count_weight_gt0(whole)
    assert np.array_equal(bits(slabs.tsdf), bits(whole.tsdf)) and np.array_equal(bits(slabs.weight), bits(whole.weight))
    t1 = whole.tsdf.copy()
    O.integrate(whole, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam)
    seen = whole.weight == 2
    assert seen.sum() == total and np.allclose(whole.tsdf[seen], t1[seen], atol=1e-6)
    mv, mn, _ = O.raycast(whole, False, pose, 0.7 * trunc, ocam, 0.3, 4.0)
    hit = mv[..., 3] != 0
    d_cam = (mv[..., :3] - pose[:3, 3]) @ pose[:3, :3]
    assert hit.sum() > 1000 and np.max(np.abs(d_cam[hit][:, 2] - tr[hit])) < 0.06        # within ~one voxel of the input depth
    thr = 300 * size / res
    all_t = O.marching_cubes(whole, False, thr, 400000)
    parts = [O.marching_cubes(whole, False, thr, 400000, z0, z0 + 16) for z0 in range(0, 64, 16)]
    cat = np.concatenate(parts)
    assert len(all_t) > 500 and np.array_equal(cat.view(np.uint32), all_t.view(np.uint32))


def test_solver_closed_form():
    """6x6 solve + Euler increment against numpy on a synthetic SPD system; exp map against scipy's matrix exponential."""
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    J = rng.normal(size=(500, 6))
    x_true = np.array([0.01, -0.02, 0.015, 0.03, -0.01, 0.02])
    r = J @ x_true
    A, b = J.T @ J, J.T @ r
    packed = []
    for i in range(6):
        for j in range(i, 7):
            packed.append(b[i] if j == 6 else A[i, j])
    ok, x = O.solve6(np.array(packed, np.float32))
    assert ok and np.allclose(x, x_true, atol=2e-5)
    ok, T = O.vector6_to_transform(x, 0.3, 0.3)
    assert ok and np.allclose(T[:3, 3], x_true[3:], atol=2e-5) and np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-6)
    ok, _ = O.vector6_to_transform(np.array([0.4, 0, 0, 0, 0, 0], np.float32), 0.3, 0.3)
    assert not ok
    ok, _ = O.solve6(np.zeros(27, np.float32))
    assert not ok                                         # det < 1e-10 -> lost
    v = np.array([0.1, -0.2, 0.05, 0.3, 0.1, -0.2])
    R, t = O.exp_map(v)
    tw = np.zeros((4, 4))
    tw[:3, :3] = [[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]]
    tw[:3, 3] = v[3:]
    E = expm(tw)
    assert np.allclose(R, E[:3, :3], atol=1e-12) and np.allclose(t, E[:3, 3], atol=1e-12)


def test_pixel_rounding_floor_form_equals_reference_double_form():
    """The packed fusion kernel rounds a projected pixel with floor(p + 0.5f) in fp32 (integrate.hip, k_integrate_pairs); the
    reference computes (int)(p + 0.5) with a double literal (src/cuda/DepthCamera.h:42).  For every p >= 0.5 below 2^23 the fp32
    sum is exact unless it crosses into the next binade, and there it can only lose a bit that does not reach the integer part."""
    rng = np.random.default_rng(0)
    parts = [rng.uniform(0.5, 2048, 2_000_000).astype(np.float32)]
    for top in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
        for centre in (np.float32(top), np.float32(top - 0.5)):
            bits = int(centre.view(np.uint32))
            parts.append((np.arange(-40000, 40000, dtype=np.int64) + bits).astype(np.uint32).view(np.float32))
    p = np.concatenate(parts)
    p = p[p >= 0.5]
    want = (p.astype(np.float64) + 0.5).astype(np.int64)
    got = np.floor((p + np.float32(0.5)).astype(np.float32)).astype(np.int64)
    assert np.array_equal(want, got)
    # below 0.5 both forms give a value < 1, which the bounds test rejects either way
    q = rng.uniform(-50, 0.5, 100000).astype(np.float32)
    assert np.all((q.astype(np.float64) + 0.5).astype(np.int64) < 1) and np.all(np.floor(q + np.float32(0.5)) < 1)


def test_float_to_int_follows_the_cuda_path():
    """(int) of NaN / out-of-range: the north star names the reference's CUDA path, whose cvt.rzi.s32 saturates and maps
    NaN to 0 (x86-64 would give INT_MIN); gfx950's v_cvt_i32 does the same natively.  DESIGN.md section 2.3."""
    assert O.to_int(float("nan")) == 0
    assert O.to_int(1e20) == 2**31 - 1 and O.to_int(float("inf")) == 2**31 - 1
    assert O.to_int(-1e20) == -2**31 and O.to_int(float("-inf")) == -2**31
    assert O.to_int(-3.99) == -3 and O.to_int(3.99) == 3 and O.to_int(2147483647.5) == 2**31 - 1


def test_color_weight_fp32_form_equals_reference_double_form():
    """integrateVolume.cu:72 forms the colour weight in double -- fminf(1.0, abs(normalz) / 0.75) * 2.0 -- and narrows it.  The HIP kernel
    uses 2.0f * fminf(1.0f, |nz| / 0.75f): the narrowed double quotient equals the correctly rounded fp32 quotient for EVERY float (the exact
    value 4|nz|/3 never comes within double rounding's reach of a float midpoint); checked over all mantissas of two binades (scaling by
    two changes nothing in the normal range), all denormals, and random values across the exponent range."""
    def mismatches(x):
        x = x.astype(np.float32)
        ref = np.minimum(np.float32(1.0), (x.astype(np.float64) / 0.75).astype(np.float32)).astype(np.float64) * 2.0
        got = np.float32(2.0) * np.minimum(np.float32(1.0), x / np.float32(0.75))
        return int(np.count_nonzero(ref.astype(np.float32).view(np.uint32) != got.astype(np.float32).view(np.uint32)))
    m = np.arange(2 ** 23, dtype=np.uint32)
    assert mismatches((m + np.uint32(0x3F000000)).view(np.float32)) == 0        # [0.5, 1)
    assert mismatches((m + np.uint32(0x3F800000)).view(np.float32)) == 0        # [1, 2)
    assert mismatches(m.view(np.float32)) == 0                                  # zero and the denormals
    rng = np.random.default_rng(5)
    assert mismatches(np.exp(rng.uniform(-80, 3, 2_000_000))) == 0


def test_z_band_volume_equals_the_same_layers_of_the_whole_volume():
    """A z-band oracle volume (layers [z_base, z_base + n) of the grid only: how 2048^3, 103 GB as a whole, gets under the oracle) must be the
    whole-volume oracle restricted to those layers: integrate counts and planes, marching cubes over an inner range, no read outside the band --
    and a read that does leave the band is counted, not answered silently."""
    cam = (160, 120, 79.5, 59.5, 131.25, 131.25)
    ocam = O.Cam.make(*cam)
    size, res, trunc = 3.0, 64, 0.1
    whole = O.OVolume(res, size, 128.0)
    band = O.OVolume(res, size, 128.0, band=(22, 28))              # layers 22 .. 49
    for k in (0, 5):
        pose = S.trajectory_pose(k, size).astype(np.float32)
        tr = O.trunc_depth(O.depth_mm_to_m(S.render_depth_mm(pose, cam, size)), 0.3, 4.0)
        n = O.vertices_to_normals(O.depth_to_vertices(tr, ocam))
        O.integrate(whole, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam)
        n_band = O.integrate(band, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam, 22, 50)
        ref = O.OVolume(res, size, 128.0)                           # the same count from a whole volume restricted to the layers
        assert n_band == O.integrate(ref, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam, 22, 50)
    assert np.array_equal(bits(band.tsdf), bits(whole.tsdf[22:50])) and np.array_equal(bits(band.weight), bits(whole.weight[22:50]))
    assert O.count_weight_gt0(band) == int((whole.weight[22:50] > 0).sum()) > 1000
    thr = 300 * size / res
    a = O.marching_cubes(whole, False, thr, 400000, 24, 48)
    b = O.marching_cubes(band, False, thr, 400000, 24, 48)         # corner interpolations reach one layer beyond the cells: inside the band
    assert len(a) > 200 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert O.band_violations() == 0
    O.marching_cubes(band, False, thr, 400000, 20, 24)             # leaves the band below: counted
    assert O.band_violations() > 0
    assert O.integrate(band, tr, n, None, False, False, pose, trunc, 2.5, ocam, ocam, 0, 64) == 0 and O.band_violations() == 1
