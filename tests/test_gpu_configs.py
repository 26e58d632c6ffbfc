"""GPU: every configuration BASELINE.json names, at its FULL size, through the C ABI against the CPU oracle, plus the whole
drop-in boundary (the twelve 1:1 wrappers of include/hybkf.h in the reference's own call order).

C1  256^3 @ 3 m, 3-level ICP, frames read by DataSourceProducerRGBDDataset from a TUM-format directory
C2  512^3 @ 4 m, VGA: raycast maps and marching-cubes triangle sequence vs the oracle (integrate is in test_gpu_parity.py)
C3  512^3 @ 4 m, VGA, CameraPoseFinderSDF: 27 sums and tracked poses vs okf_sdf_* on identical inputs
C4  1024^3 @ 6 m, depth gates opened: update counts + planes vs the oracle, and 2 z-slabs == whole volume
C5  2048^3 @ 8 m, 1280x960: three z-bands of the 68.7 GB volume against a z-band oracle (update counts, planes, marching cubes), voxel
    indices beyond 2^31 and 2^32 (the tracked pipeline at this size: test_gpu_parity.py::test_maximum_configuration_2048_cubed)
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from hybkinectfu_amd import host_app as H
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import pipeline as PL
from hybkinectfu_amd import scene as S
from hybkinectfu_amd import tum

pytestmark = pytest.mark.gpu
P = S.STOCK


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def oracle_preprocess(mm, ocam, tmax=P["depth_trunc_max"]):
    tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], tmax)
    fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
    v = O.depth_to_vertices(fl, ocam)
    return tr, fl, v, O.vertices_to_normals(v)


def host_ram_gib():
    import psutil
    return psutil.virtual_memory().available / 2**30


# ---------------------------------------------------------------------------------------------------------------------------
# the drop-in boundary: src/HybKinectfu.cpp:106-110 + CameraPoseFinderICP.cpp:57-60 call these twelve wrappers one by one
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sigma_pixel", [2.0, 1.5, 2.5])          # radius = ceil(2 sigma) = 4 (fused path), 3 and 5 (generic kernel)
def test_per_call_wrappers_equal_fused_preprocess(sigma_pixel):
    """kf_trunc_depth -> kf_bilateral_filter_depth -> kf_calculate_new_vertices -> kf_calculate_new_normals -> 4 x kf_downsample_*
    (what INTEGRATION.md section A forwards the reference's cuda* wrappers to) == kf_preprocess + the tracker's own pyramid
    launch, bit for bit on every map; and both == the oracle (bilateral: 2e-6 relative, device __expf)."""
    cam, size = S.vga_camera(), 4.0
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    mm = S.render_depth_mm(S.trajectory_pose(11, size), cam, size)
    mm[100:140, 200:260] = 0
    mm[300, :] = 50000
    a = K.Context(kcam, 64, size, levels=3)
    b = K.Context(kcam, 64, size, levels=3)
    a.upload_depth_mm(mm)                                                 # copyFrameToGPU
    a.trunc_depth(P["depth_trunc_min"], P["depth_trunc_max"])             # cudaTruncDepth
    a.bilateral(sigma_pixel, P["filter_sigma_depth"])                     # cudaBiliteralFilterDepth
    a.calculate_new_vertices()                                            # cudaCalculateNewVertices
    a.calculate_new_normals()                                             # cudaCalculateNewNormals
    a.downsample(model=False)                                             # cudaDownSampleNewVertices / NewNormals
    b.upload_depth_mm(mm)
    b.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], sigma_pixel, P["filter_sigma_depth"])
    b.downsample(model=False)
    for m in (K.MAP_RAW_DEPTH, K.MAP_TRUNCED_DEPTH, K.MAP_FILTERED_DEPTH):
        assert np.array_equal(bits(a.download_map(m)), bits(b.download_map(m))), m
    for m in (K.MAP_NEW_VERTICES, K.MAP_NEW_NORMALS):
        for lvl in range(3):
            assert np.array_equal(bits(a.download_map(m, lvl)), bits(b.download_map(m, lvl))), (m, lvl)
    # against the oracle
    tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
    fl = O.bilateral(tr, sigma_pixel, P["filter_sigma_depth"])
    g_fl = a.download_map(K.MAP_FILTERED_DEPTH)
    assert np.array_equal(bits(a.download_map(K.MAP_TRUNCED_DEPTH)), bits(tr))
    assert np.allclose(g_fl, fl, rtol=2e-6, atol=0) and np.array_equal(g_fl == 0, fl == 0)
    v = O.depth_to_vertices(g_fl, ocam)
    n = O.vertices_to_normals(v)
    for lvl, (ov, on) in enumerate(zip(O.pyramid(v, 3), O.pyramid(n, 3, normals=True))):
        assert np.array_equal(bits(a.download_map(K.MAP_NEW_VERTICES, lvl)), bits(ov))
        assert np.array_equal(bits(a.download_map(K.MAP_NEW_NORMALS, lvl)), bits(on))
    # model side: cudaDownSampleModelVertices / ModelNormals on uploaded raycast maps
    rng = np.random.default_rng(3)
    mv = rng.standard_normal((cam[1], cam[0], 4)).astype(np.float32); mv[..., 3] = 1
    mn = rng.standard_normal((cam[1], cam[0], 4)).astype(np.float32); mn[..., 3] = 0
    mv[50:60, 70:90] = 0; mn[50:60, 70:90] = 0
    a.upload_map(K.MAP_MODEL_VERTICES, 0, mv); a.upload_map(K.MAP_MODEL_NORMALS, 0, mn)
    a.downsample(model=True)
    for lvl, (ov, on) in enumerate(zip(O.pyramid(mv, 3), O.pyramid(mn, 3, normals=True))):
        assert np.array_equal(bits(a.download_map(K.MAP_MODEL_VERTICES, lvl)), bits(ov))
        assert np.array_equal(bits(a.download_map(K.MAP_MODEL_NORMALS, lvl)), bits(on))
    a.close(); b.close()


def test_bilateral_radius_beyond_kernel_limit_is_refused():
    ctx = K.Context(K.camera(64, 48, 31.5, 23.5, 52.5, 52.5), 32, 3.0, levels=3)
    ctx.upload_depth_mm(np.full((48, 64), 1000, np.uint16))
    ctx.trunc_depth(0.3, 4.0)
    with pytest.raises(K.KfError):
        ctx.bilateral(5.0, 0.03)                                   # radius 10 > 8: KF_ERR_ARG, never a silent clamp
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------------
# C1
# ---------------------------------------------------------------------------------------------------------------------------
def test_c1_dataset_reader_256_cubed(tmp_path):
    """BASELINE configs[0]: 256^3 @ 3 m, VGA, 3-level ICP, frames from a TUM-format directory through
    DataSourceProducerRGBDDataset (the real fr1_xyz sequence is not in the image: Scene S written in its format, 5000 units/m).
    The fused volume equals the oracle's bit for bit when the oracle integrates with the poses the device tracked."""
    res, size, cam, n = 256, 3.0, S.vga_camera(), 5
    ocam = O.Cam.make(*cam)
    frames = [S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)]
    d = str(tmp_path / "seq") + "/"
    tum.write_dataset(d, frames, poses=[S.trajectory_pose(k, size) for k in range(n)])
    app = H.App(res, size, cam, dataset_dir=d)                      # stock sdf truncation 0.05 m = 4.3 voxels
    ctx = K.Context.borrow(app.ctx_handle(), K.camera(*cam), res, size)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    for k in range(n):
        r = app.process_dataset_frame(k)
        assert r is not None and r[0], k
        tracked, pose = app.pose()
        assert tracked and np.linalg.norm(pose[:3, 3] - S.trajectory_pose(k, size)[:3, 3]) < 6e-3
        # the reader hands over exactly the millimetres that were written (PNG decode + /5 rounding)
        assert np.array_equal(bits(ctx.download_map(K.MAP_RAW_DEPTH)), bits(O.depth_mm_to_m(frames[k])))
        tr, fl, v, nn = oracle_preprocess(frames[k], ocam)
        n_o = O.integrate(ovol, tr, nn, None, False, False, pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"], ocam, ocam)
        assert ctx.stats()["updated_last"] == n_o and n_o > 100000
    t, w = ctx.download_volume()
    assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    assert ctx.stats()["weight_gt0"] == O.count_weight_gt0(ovol)
    app.close()


# ---------------------------------------------------------------------------------------------------------------------------
# C2: raycast + marching cubes at full size
# ---------------------------------------------------------------------------------------------------------------------------
def test_c2_raycast_and_marching_cubes_512_cubed():
    res, size, cam = 512, 4.0, S.vga_camera()
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=3_000_000)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    pose = None
    for k in (0, 4, 8):
        pose = S.trajectory_pose(k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        tr, fl, v, n = oracle_preprocess(mm, ocam)
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"], ocam, ocam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ctx.integrate(pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"])
        assert ctx.stats()["updated_last"] == n_o
    inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
    ov, on, _ = O.raycast(ovol, False, pose, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    ctx.raycast(pose, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    gv, gn = ctx.download_map(K.MAP_MODEL_VERTICES), ctx.download_map(K.MAP_MODEL_NORMALS)
    assert int((ov[..., 3] != 0).sum()) > 100000
    assert np.array_equal(bits(gv), bits(ov)) and np.array_equal(bits(gn), bits(on))
    # a second view, far off the fused poses: grazing rays, rays leaving through the side faces
    pose2 = S.trajectory_pose(37, size).astype(np.float32)
    ov, on, _ = O.raycast(ovol, False, pose2, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"])
    ctx.raycast(pose2, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_VERTICES)), bits(ov))
    assert np.array_equal(bits(ctx.download_map(K.MAP_MODEL_NORMALS)), bits(on))
    thr = 300 * size / res
    otris = O.marching_cubes(ovol, False, thr, 3_000_000)
    ctx.marching_cubes(thr)
    g = ctx.triangles()
    assert len(otris) > 100000 and len(g) == len(otris)
    assert np.array_equal(g["v"]["pos"].view(np.uint32), otris["v"]["pos"].view(np.uint32))       # same triangles, same order
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------------
# C3: SDF tracker at 512^3
# ---------------------------------------------------------------------------------------------------------------------------
def test_c3_sdf_tracker_512_cubed():
    """CameraPoseFinderSDF (src/CameraPoseFinderSDF.cpp:44-106) at 512^3 @ 4 m, VGA: per frame the 27 sums (<= 1e-5 of the
    largest entry vs fp64 accumulation) and the tracked pose (<= 1e-4, same iteration count and verdict) against the oracle on
    IDENTICAL inputs: both volumes are fused with the oracle's poses and stay bit-identical."""
    res, size, cam = 512, 4.0, S.vga_camera()
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    pose = S.pose0(size)
    ctx.set_pose(pose)
    tracked_frames = 0
    for k in range(5):
        mm = S.render_depth_mm(S.trajectory_pose(k, size), cam, size)
        tr, fl, v, n = oracle_preprocess(mm, ocam)
        ctx.upload_depth_mm(mm)
        ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        assert np.array_equal(bits(ctx.download_map(K.MAP_TRUNCED_DEPTH)), bits(tr))
        if k > 0:
            sd, sf, valid = O.sdf_system(ovol, tr, ocam, pose)
            g = ctx.sdf_system(pose)
            assert valid > 20000
            assert np.max(np.abs(g - sd)) <= 1e-5 * np.max(np.abs(sd)), (k, g, sd)
            ok_o, pose_o, it_o = O.sdf_estimate(ovol, tr, ocam, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
            ctx.set_pose(pose)
            ctx.sdf_track(k, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])
            ok_g, pose_g, status, iters = ctx.track_result()
            assert ok_g == ok_o, (k, status)
            if ok_o:
                tracked_frames += 1
                assert iters == it_o
                assert np.max(np.abs(pose_g[:3, 3] - pose_o[:3, 3])) < 1e-4 and np.max(np.abs(pose_g[:3, :3] - pose_o[:3, :3])) < 1e-4, (k, pose_g, pose_o)
                pose = np.array(pose_o, np.float32)
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"], ocam, ocam)
        ctx.integrate(pose, P["integrate_sdf_trunc"], P["integrate_depth_trunc"])
        assert ctx.stats()["updated_last"] == n_o
    assert tracked_frames >= 3
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------------
# C4: 1024^3 @ 6 m
# ---------------------------------------------------------------------------------------------------------------------------
def test_c4_integrate_1024_cubed_and_two_slabs():
    if host_ram_gib() < 24:
        pytest.skip("the oracle's 1024^3 volume needs 12.9 GB of host memory")
    res, size, cam = 1024, 6.0, S.vga_camera()
    tmax = dist = 6.0                                              # C4: gates raised to the volume size
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
    halo = PL.slab_halo_layers(res, size, inc)
    assert halo == 8
    whole = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=4_000_000)
    slabs = [K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=4_000_000, slab=r, halo=halo) for r in PL.slab_ranges(res, 2)]
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    import torch
    dev = torch.device("cuda", 0)
    bufs = [(torch.empty((cam[1], cam[0]), dtype=torch.float32, device=dev), torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev),
             torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev)) for _ in slabs]
    pose = None
    for k in (0, 6):
        pose = S.trajectory_pose(k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        tr, fl, v, n = oracle_preprocess(mm, ocam, tmax)
        n_o = O.integrate(ovol, tr, n, None, False, False, pose, P["integrate_sdf_trunc"], dist, ocam, ocam)
        for c in [whole] + slabs:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], tmax, P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.integrate(pose, P["integrate_sdf_trunc"], dist)
        st = whole.stats()
        assert st["updated_last"] == n_o and n_o > 30_000_000
        assert st["weight_gt0"] == O.count_weight_gt0(ovol)
        assert sum(c.stats()["weight_gt0"] for c in slabs) == st["weight_gt0"]
    # planes: a z range in front of the camera and one across the slab boundary at z = 512
    for z0, z1 in ((96, 160), (480, 544)):
        t, w = whole.download_volume(z0, z1)
        assert np.array_equal(bits(t), bits(ovol.tsdf[z0:z1])) and np.array_equal(bits(w), bits(ovol.weight[z0:z1]))
    for c, (z0, z1) in zip(slabs, PL.slab_ranges(res, 2)):
        a, b = (z1 - 64, z1) if z0 == 0 else (z0, z0 + 64)
        t, w = c.download_volume(a, b)
        assert np.array_equal(bits(t), bits(ovol.tsdf[a:b])) and np.array_equal(bits(w), bits(ovol.weight[a:b]))
    # raycast: whole == oracle, merged slabs == whole (first crossing along the ray wins)
    ov, on, _ = O.raycast(ovol, False, pose, inc, ocam, P["depth_trunc_min"], tmax)
    whole.raycast(pose, inc, P["depth_trunc_min"], tmax)
    wv, wn = whole.download_map(K.MAP_MODEL_VERTICES), whole.download_map(K.MAP_MODEL_NORMALS)
    assert int((ov[..., 3] != 0).sum()) > 100000
    assert np.array_equal(bits(wv), bits(ov)) and np.array_equal(bits(wn), bits(on))
    for c, (t, v, n) in zip(slabs, bufs):
        c.raycast_slab(pose, inc, P["depth_trunc_min"], tmax, t.data_ptr(), v.data_ptr(), n.data_ptr())
        c.sync()
    tmin = torch.minimum(bufs[0][0], bufs[1][0])
    merged = []
    for which in (1, 2):
        acc = torch.zeros((cam[1], cam[0], 4), dtype=torch.int32, device=dev)
        for b in bufs:
            win = (b[0] == tmin) & torch.isfinite(b[0])
            acc += b[which].view(torch.int32) * win.unsqueeze(-1).to(torch.int32)
        merged.append(acc.view(torch.float32).cpu().numpy())
    assert np.array_equal(bits(merged[0]), bits(wv)) and np.array_equal(bits(merged[1]), bits(wn))
    assert all(int(torch.isfinite(b[0]).sum()) > 1000 for b in bufs)        # both slabs really contribute crossings
    for c in [whole] + slabs:
        c.close()


# ---------------------------------------------------------------------------------------------------------------------------
# C5: 2048^3 @ 8 m, 1280x960 -- the volume is 103 GB in the oracle's layout, so the oracle holds one z-band of it at a time
# ---------------------------------------------------------------------------------------------------------------------------
def test_c5_2048_cubed_z_bands_against_the_oracle():
    """BASELINE.json's largest configuration, bit for bit where the oracle can reach it: three 64-layer bands of the whole 2048^3 volume
    (free space in front of the room; the central sphere's cap, voxel indices beyond 2^31; the back wall, beyond 2^32) after two fused
    frames -- update count of the band per frame (a slab context of exactly those layers), tsdf / weight planes downloaded from the
    WHOLE-volume context (64-bit voxel addressing, > 1 M-entry brick queue, the 12288-workgroup grid, deferred free-space weights at
    scale), and the marching-cubes triangle sequence of an inner range (slab extraction).  Reference: integrateVolume.cu:15-77,
    tsdfVolume.h:57-60, marchingcube.cu:41-152."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2**30:
        pytest.skip("needs ~80 GB of free HBM")
    if host_ram_gib() < 10:
        pytest.skip("a 64-layer band of the oracle's 2048^3 volume needs 3.2 GB of host memory")
    res, size, cam = 2048, 8.0, S.vga_camera(2)
    tmax = dist = 8.0
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    maxw, trunc = P["volume_max_weight"], P["integrate_sdf_trunc"]
    thr = 300 * size / res
    # (band layers, marching-cubes layers inside them): 1.25-1.5 m free space | sphere cap at 2.8 m | back wall at 6 m
    bands = [((320, 384), None), ((704, 768), (712, 728)), ((1504, 1568), (1528, 1544))]
    whole = K.Context(kcam, res, size, maxw, levels=3)
    slab = [K.Context(kcam, res, size, maxw, levels=3, slab=b) for b, _ in bands]
    mcs = [K.Context(kcam, res, size, maxw, levels=3, slab=m, halo=8, max_triangles=3_000_000) if m else None for _, m in bands]
    frames, n_slab = [], []
    for k in (0, 6):
        pose = S.trajectory_pose(k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        tr, fl, v, n = oracle_preprocess(mm, ocam, tmax)
        frames.append((pose, tr, n))
        for c in [whole] + slab + [m for m in mcs if m]:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], tmax, P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.integrate(pose, trunc, dist)
        n_slab.append([c.stats()["updated_last"] for c in slab])
    st = whole.stats()
    assert st["updated_last"] > 4e8 and st["bricks_active"] > 100_000        # (second frame: whole free-space bricks are retired by the cull, not queued)
    for i, ((z0, z1), mz) in enumerate(bands):
        ovol = O.OVolume(res, size, maxw, band=(z0, z1 - z0))
        for f, (pose, tr, n) in enumerate(frames):
            n_o = O.integrate(ovol, tr, n, None, False, False, pose, trunc, dist, ocam, ocam, z0, z1)
            assert n_o == n_slab[f][i] and n_o > 1_000_000, (i, f, n_o, n_slab[f][i])
        t, w = whole.download_volume(z0, z1)                       # voxel index (z * 2048 + y) * 2048 + x of the 2048^3 grid: up to 6.6e9
        assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight)), (z0, z1)
        t, w = slab[i].download_volume(z0, z1)
        assert np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight)), (z0, z1)
        assert int((ovol.weight == 2).sum()) > 500_000              # both frames really landed in the band
        if mz:
            ot = O.marching_cubes(ovol, False, thr, 3_000_000, mz[0], mz[1])
            mcs[i].marching_cubes(thr)
            gt = mcs[i].triangles()
            assert len(ot) > 10_000, (mz, len(ot))
            assert len(gt) == len(ot) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)), (mz, len(gt), len(ot))
        assert O.band_violations() == 0
        del ovol
    for c in [whole] + slab + [m for m in mcs if m]:
        c.close()
