"""GPU (one device): N z-slab contexts, merged exactly as SlabPipeline merges them over RCCL, reproduce the whole-volume
context bit for bit -- TSDF planes, update counts, raycast maps, tracked pose, marching-cubes triangle sequence."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import pipeline as PL
from hybkinectfu_amd import scene as S

pytestmark = pytest.mark.gpu
P = S.STOCK


def _normals_both_forms(c, cam, inc, ta, ta_min, dev):
    """kf_slab_ray_normals and its speculative form (kf_raycast_volume_slab_cross_spec + kf_slab_ray_normals_spec, what SlabPipeline runs) on one context:
    the same crossing words, the same candidates, bit for bit.  Returns the candidates."""
    import torch
    ta2, own = torch.empty_like(ta), torch.empty_like(ta)
    spec = torch.empty((cam[1], cam[0], 3), dtype=torch.float32, device=dev)
    c.raycast_slab_cross_spec(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta2.data_ptr(), own.data_ptr(), spec.data_ptr())
    c.sync()
    assert torch.equal(ta2, ta) and torch.equal(own, ta)
    cand = torch.empty((cam[1], cam[0], 3), dtype=torch.float32, device=dev)
    c.slab_ray_normals(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta_min.data_ptr(), cand.data_ptr())
    cand2 = torch.full((cam[1], cam[0], 3), 7.0, dtype=torch.float32, device=dev)
    c.slab_ray_normals_spec(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta_min.data_ptr(), own.data_ptr(), spec.data_ptr(), cand2.data_ptr())
    c.sync()
    assert torch.equal(cand.view(torch.int32), cand2.view(torch.int32))
    # where this context's own crossing won, the candidate IS the speculative gradient
    won = (own == ta_min) & ((ta_min & 0xFFFFFFFF) != 0)
    assert torch.equal(cand2.view(torch.int32)[won], spec.view(torch.int32)[won])
    return cand



@pytest.mark.parametrize("world,maxw,frames,balanced", [(2, P["volume_max_weight"], 3, False), (4, P["volume_max_weight"], 3, False),
                                                        (2, 3.0, 9, False),       # max_weight 3: from frame 3 on the saturation-aware fusion runs in every context
                                                        (3, P["volume_max_weight"], 3, True)])   # unequal slabs from the one-frame work probe (bench.py's default for N > 1)
def test_slabs_equal_whole_volume(world, maxw, frames, balanced):
    cam = S.vga_camera()
    kcam = K.camera(*cam)
    size, res = 3.0, 128
    trunc = 5 * size / res
    inc = 0.7 * trunc
    halo = PL.slab_halo_layers(res, size, inc)
    ranges = PL.slab_ranges(res, world)
    if balanced:
        first = torch.from_numpy(S.render_depth_mm(S.trajectory_pose(0, size), cam, size).astype(np.int16)).cuda()
        work = PL.probe_layer_work(kcam, res, size, None, first.data_ptr(), probe_res=64)
        ranges = PL.slab_ranges(res, world, work, halo=halo)
        assert ranges != PL.slab_ranges(res, world) and ranges[0][0] == 0 and ranges[-1][1] == res      # the probe moved the boundaries
        assert 1.0e4 < sum(work) < 2.1e6 and len(work) == res // 8                                      # a plausible voxel count per frame (128^3 holds 2.1 M voxels)
    whole = K.Context(kcam, res, size, maxw, levels=3, max_triangles=600000)
    slabs = [K.Context(kcam, res, size, maxw, levels=3, max_triangles=600000, slab=r, halo=halo) for r in ranges]
    dev = torch.device("cuda", 0)
    bufs = [(torch.empty((cam[1], cam[0]), dtype=torch.float32, device=dev), torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev),
             torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev)) for _ in slabs]
    pose = S.pose0(size)
    for c in [whole] + slabs:
        c.set_pose(pose)
    for k in range(frames):
        mm = S.render_depth_mm(S.trajectory_pose(k, size), cam, size)
        for c in [whole] + slabs:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
            c.integrate(None, trunc, 2.5)
        ok_w, pose_w, _, _ = whole.track_result()
        assert ok_w
        for c in slabs:                                           # replicated tracking: identical bits on every rank
            ok_s, pose_s, _, _ = c.track_result()
            assert ok_s and np.array_equal(pose_s.view(np.uint32), pose_w.view(np.uint32))
        whole.raycast(None, inc, P["depth_trunc_min"], P["depth_trunc_max"])
        for c, (t, v, n) in zip(slabs, bufs):
            c.raycast_slab(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], t.data_ptr(), v.data_ptr(), n.data_ptr())
            c.sync()
        # the collective, emulated in-process: MIN over ranks, then integer SUM of the masked candidates
        ts = torch.stack([b[0] for b in bufs])
        tmin = ts.min(dim=0).values
        merged = []
        for which in (1, 2):
            acc = torch.zeros((cam[1], cam[0], 4), dtype=torch.int32, device=dev)
            for b in bufs:
                win = (b[0] == tmin) & torch.isfinite(b[0])
                acc += b[which].view(torch.int32) * win.unsqueeze(-1).to(torch.int32)
            merged.append(acc.view(torch.float32).contiguous())
        wv, wn = whole.download_map(K.MAP_MODEL_VERTICES), whole.download_map(K.MAP_MODEL_NORMALS)
        assert int((wv[..., 3] != 0).sum()) > 10000
        assert np.array_equal(merged[0].cpu().numpy().view(np.uint32), wv.view(np.uint32))
        assert np.array_equal(merged[1].cpu().numpy().view(np.uint32), wn.view(np.uint32))
        # the merge SlabPipeline runs: 64-bit crossing words (MIN), normals by the vertex's owner (integer SUM), vertices rebuilt from the rays --
        # must arrive at the same bits as the whole-volume maps
        tas = []
        for c in slabs:
            ta = torch.empty((cam[1], cam[0]), dtype=torch.int64, device=dev)
            c.raycast_slab_cross(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta.data_ptr())
            c.sync()
            tas.append(ta)
        ta_min = torch.stack(tas).min(dim=0).values.contiguous()
        acc = torch.zeros((cam[1], cam[0], 3), dtype=torch.int32, device=dev)
        owners = torch.zeros((cam[1], cam[0]), dtype=torch.int32, device=dev)
        for c, ta in zip(slabs, tas):
            cand = _normals_both_forms(c, cam, inc, ta, ta_min, dev)
            acc += cand.view(torch.int32)
            owners += (cand.view(torch.int32) != 0).any(dim=-1).to(torch.int32)
        rays = acc.view(torch.float32).contiguous()
        assert int(owners.max()) == 1                                                      # exactly one owner per vertex
        for c in slabs:
            c.set_model_maps_rays(None, ta_min.data_ptr(), rays.data_ptr())
            c.sync()
            assert np.array_equal(c.download_map(K.MAP_MODEL_VERTICES).view(np.uint32), wv.view(np.uint32))
            assert np.array_equal(c.download_map(K.MAP_MODEL_NORMALS).view(np.uint32), wn.view(np.uint32))
        for c in slabs[:1]:                                        # (and the map form's hand-over still works)
            c.set_model_maps_device(merged[0].data_ptr(), merged[1].data_ptr())
            c.sync()
    # volumes: owned layers of every slab equal the whole volume; update counts add up over the owned layers
    tw, ww = whole.download_volume()
    for c, (z0, z1) in zip(slabs, ranges):
        t, w = c.download_volume(z0, z1)
        assert np.array_equal(t.view(np.uint32), tw[z0:z1].view(np.uint32)) and np.array_equal(w, ww[z0:z1])
    assert sum(c.stats()["weight_gt0"] for c in slabs) == whole.stats()["weight_gt0"]
    # marching cubes: slab-major concatenation == whole-volume sequence
    thr = 300 * size / res
    whole.marching_cubes(thr)
    wt = whole.triangles()
    parts = []
    for c in slabs:
        c.marching_cubes(thr)
        parts.append(c.triangles())
    cat = np.concatenate(parts)
    assert len(wt) > 1000 and len(cat) == len(wt) and np.array_equal(cat.view(np.uint32), wt.view(np.uint32))
    for c in [whole] + slabs:
        c.close()


def test_pixel_partitioned_icp_matches_replicated():
    """Two contexts play two ranks: each sums half of the image rows per Gauss-Newton step, the 27-float systems are added
    (the all-reduce) and both apply the same update.  Result: same verdict, pose within 1e-4 of the replicated tracker."""
    cam = S.vga_camera()
    kcam = K.camera(*cam)
    size, res = 3.0, 128
    trunc = 5 * size / res
    ref = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    ranks = [K.Context(kcam, res, size, P["volume_max_weight"], levels=3) for _ in range(2)]
    dev = torch.device("cuda", 0)
    sums = [torch.zeros(32, dtype=torch.float32, device=dev) for _ in ranks]
    for c in [ref] + ranks:
        c.set_pose(S.pose0(size))
    for k in range(3):
        mm = S.render_depth_mm(S.trajectory_pose(k, size), cam, size)
        for c in [ref] + ranks:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        ref.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        icp = K.IcpParams(3, P["icp_thre_sin_angle"], P["icp_thre_dist"], P["camera_shake_dist"], P["camera_shake_angle"])
        lib = K.load()
        for r, c in enumerate(ranks):
            assert lib.kf_icp_partition_begin(c.h, k) == 0
        if k > 0:
            import ctypes as C
            for step in range(lib.kf_icp_partition_steps(ranks[0].h)):
                for r, c in enumerate(ranks):
                    assert lib.kf_icp_partition_step(c.h, step, C.byref(icp), C.byref(c.cam), r, 2, C.c_void_p(sums[r].data_ptr())) == 0
                    c.sync()
                total = sums[0] + sums[1]                               # the all-reduce
                for s_ in sums:
                    s_.copy_(total)
                torch.cuda.synchronize()
            for r, c in enumerate(ranks):
                assert lib.kf_icp_partition_finish(c.h, C.byref(icp), C.c_void_p(sums[r].data_ptr())) == 0
        ok_ref, pose_ref, _, it_ref = ref.track_result()
        res_r = [c.track_result() for c in ranks]
        assert ok_ref and all(r[0] for r in res_r)
        assert np.array_equal(res_r[0][1].view(np.uint32), res_r[1][1].view(np.uint32))        # ranks agree bitwise
        assert np.max(np.abs(res_r[0][1] - pose_ref)) < 1e-4 and res_r[0][3] == it_ref
        for c in [ref] + ranks:
            c.integrate(None, trunc, 2.5)
            c.raycast(None, 0.7 * trunc, P["depth_trunc_min"], P["depth_trunc_max"])
    for c in [ref] + ranks:
        c.close()


def test_device_mask_equals_torch_merge_rule():
    """kf_slab_mask_candidates (one launch) == the masking step of pipeline.merge_candidates (plain torch), bit for bit,
    including -0.0 components, +inf 'no crossing' entries and ties between equal finite parameters."""
    import torch
    from hybkinectfu_amd import pipeline as PL
    cam = (64, 48, 31.5, 23.5, 52.5, 52.5)
    ctx = K.Context(K.camera(*cam), 32, 3.0, levels=3)
    g = torch.Generator().manual_seed(5)
    t = torch.rand((48, 64), generator=g) * 3 + 0.5
    t[torch.rand((48, 64), generator=g) < 0.3] = float("inf")
    tmin = torch.minimum(t, torch.rand((48, 64), generator=g) * 3 + 0.5)
    tmin[5, :] = t[5, :]                                              # rows where this rank wins outright
    v = torch.randn((48, 64, 4), generator=g); n = torch.randn((48, 64, 4), generator=g)
    v[7, 3, 1] = -0.0; n[5, 9, 2] = -0.0
    td, tmd, vd, nd = t.cuda(), tmin.cuda(), v.cuda(), n.cuda()
    ctx.slab_mask_candidates(td.data_ptr(), tmd.data_ptr(), vd.data_ptr(), nd.data_ptr())
    ctx.sync()
    win = (t == tmin) & torch.isfinite(t)
    want_v = v.view(torch.int32) * win.unsqueeze(-1).to(torch.int32)
    want_n = n.view(torch.int32) * win.unsqueeze(-1).to(torch.int32)
    assert torch.equal(vd.cpu().view(torch.int32), want_v) and torch.equal(nd.cpu().view(torch.int32), want_n)
    assert win.any() and not win.all()
    ctx.close()


def test_slab_pipeline_over_rccl_matches_single_gpu_pipeline():
    """The pipeline bench.py runs on N > 1 GPUs (SlabPipeline: z-slab context, RCCL all-reduces, device-side masking, everything
    ordered on ONE torch stream) on a one-rank RCCL group must reproduce the plain single-GPU pipeline bit for bit: poses, model
    maps and volume after several frames.  A mis-ordered collective or kernel shows up here as a mismatch."""
    import os
    import torch.distributed as dist
    res, size, cam = 384, 3.0, S.vga_camera()                     # stock truncation (0.05 m) = 6.4 voxels, as at C2
    wl = dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"])
    n = 6
    frames = np.stack([S.render_depth_mm(S.trajectory_pose(k, size), cam, size) for k in range(n)])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        for cls in (PL.SingleGpuPipeline, PL.SlabPipeline):
            pipe = cls(K.camera(*cam), res, size, wl)
            poses = []
            for k in range(n):
                # every other frame announces its successor: the slab pipeline then preprocesses it during its merge
                nxt = dev.data_ptr() + (k + 1) * fb if (k % 2 == 0 and k + 1 < n) else None
                pipe.process_frame_device(dev.data_ptr() + k * fb, k, nxt)
                ok, pose, status, iters = pipe.track_result()
                assert ok
                poses.append(pose.copy())
            pipe.sync()
            maps = [pipe.ctx.download_map(m) for m in (K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)]
            vol = pipe.ctx.download_volume()
            outs.append((poses, maps, vol))
            pipe.close()
        for a, b in zip(outs[0][0], outs[1][0]):
            assert np.array_equal(a, b)
        for a, b in zip(outs[0][1], outs[1][1]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(outs[0][2][0], outs[1][2][0]) and np.array_equal(outs[0][2][1], outs[1][2][1])
    finally:
        dist.destroy_process_group()


def test_unpack_kernel_equals_cpu_restatement():
    """k_slab_rays_unpack (what SlabPipeline launches behind its second all-reduce) == tests/slab_cpu_ops.py, the restatement the CPU-only gloo tests
    run in its place -- bit for bit on a synthetic frame (far-extrapolated vertices, given-up crossings, failed gradients, signed zeros); the vertex
    rebuilt from its ray parameter included (rc_pixel_ray operation for operation in numpy fp32)."""
    import slab_cpu_ops as ops
    ctx = K.Context(K.camera(*ops.CAM), 32, 3.0, levels=3)
    world = 3
    tas = [ops.synthetic_crossings(48, 64, r, world, seed=11) for r in range(world)]
    ta_min = torch.stack([x[0] for x in tas]).min(dim=0).values.contiguous()
    _, _, want_ta, merged, want_v, want_n = tas[0]
    assert torch.equal(ta_min, want_ta)
    total = torch.zeros((48, 64, 3), dtype=torch.int32)
    for r in range(world):                                        # every rank's normals step (also checks the winners it is handed), integer-summed
        cand = torch.empty((48, 64, 3))
        tas[r][1](ta_min, cand)
        total += cand.view(torch.int32)
    assert torch.equal(total, merged.view(torch.int32))
    tad, md = ta_min.cuda(), merged.cuda()
    ctx.set_model_maps_rays(ops.POSE, tad.data_ptr(), md.data_ptr())
    ctx.sync()
    uv, un = ops.unpack(ta_min, merged)
    assert np.array_equal(ctx.download_map(K.MAP_MODEL_VERTICES).view(np.uint32), uv.numpy().view(np.uint32))
    assert np.array_equal(ctx.download_map(K.MAP_MODEL_NORMALS).view(np.uint32), un.numpy().view(np.uint32))
    assert torch.equal(uv.view(torch.int32), want_v.view(torch.int32)) and torch.equal(un.view(torch.int32), want_n.view(torch.int32))
    assert int((uv[..., 3] == 1).sum()) > 1000
    # the unpack launch also leaves levels 1 and 2 of the model maps' pyramids (32x8 tiles, tile-local): the same bits as the pyramid kernel's
    mine = [ctx.download_map(m, level) for level in (1, 2) for m in (K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)]
    ctx.downsample(True)                                         # k_pyramid over the same level 0
    ctx.sync()
    ref = [ctx.download_map(m, level) for level in (1, 2) for m in (K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)]
    for a, b in zip(mine, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert mine[0].any() and mine[1].any()                       # (a quarter of the synthetic pixels is empty: level 2 has no complete 4x4 block)
    ctx.close()


def test_slab_raycast_refuses_thin_halo():
    """A halo thinner than ceil(inc/voxel)+2 layers would silently lose crossings at the slab faces: KF_ERR_ARG instead."""
    cam = (64, 48, 31.5, 23.5, 52.5, 52.5)
    res, size = 128, 3.0
    dev = torch.device("cuda", 0)
    t = torch.empty((48, 64), dtype=torch.float32, device=dev)
    v = torch.empty((48, 64, 4), dtype=torch.float32, device=dev)
    n = torch.empty_like(v)
    inc = 10.5 * size / res                                        # 10.5 voxels per step: needs 11 + 2 = 13 layers
    assert PL.slab_halo_layers(res, size, inc) == 16
    thin = K.Context(K.camera(*cam), res, size, levels=3, slab=(32, 64), halo=8)
    with pytest.raises(K.KfError):
        thin.raycast_slab(S.pose0(size), inc, 0.3, 4.0, t.data_ptr(), v.data_ptr(), n.data_ptr())
    thin.close()
    ok = K.Context(K.camera(*cam), res, size, levels=3, slab=(32, 64), halo=16)
    ok.raycast_slab(S.pose0(size), inc, 0.3, 4.0, t.data_ptr(), v.data_ptr(), n.data_ptr())
    ok.sync()
    ok.close()
    # a slab touching the volume's own face needs no halo there
    edge = K.Context(K.camera(*cam), res, size, levels=3, slab=(0, 64), halo=16)
    edge.raycast_slab(S.pose0(size), inc, 0.3, 4.0, t.data_ptr(), v.data_ptr(), n.data_ptr())
    edge.sync()
    edge.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sdf_tracker_on_slabs_matches_whole_volume_and_oracle(world):
    """CameraPoseFinderSDF with the volume split into z-slabs (kf_sdf_partition_*): every context sums the pixels whose world point
    it owns, the 27-float systems are added per iteration (the all-reduce) and every context applies the sum.  The partition changes
    only the association of the fp32 sums: same verdict and iteration count as the whole-volume tracker and as the oracle, pose within
    the north star's 1e-4; the contexts agree bitwise with each other; the pixels are partitioned exactly (valid counts add up)."""
    import ctypes as C
    cam = S.vga_camera()
    kcam, ocam = K.camera(*cam), O.Cam.make(*cam)
    size, res = 3.0, 128
    trunc = 5 * size / res
    inc = 0.7 * trunc
    halo = PL.slab_halo_layers(res, size, inc)
    whole = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    slabs = [K.Context(kcam, res, size, P["volume_max_weight"], levels=3, slab=r, halo=halo) for r in PL.slab_ranges(res, world)]
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    dev = torch.device("cuda", 0)
    sums = [torch.zeros(32, dtype=torch.float32, device=dev) for _ in slabs]
    lib = K.load()
    sp = K.SdfTrackerParams(P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])
    pose = S.pose0(size)
    for c in [whole] + slabs:
        c.set_pose(pose)
    o_pose = pose.copy()
    for k in range(4):
        mm = S.render_depth_mm(S.trajectory_pose(k, size), cam, size)
        tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
        nn = O.vertices_to_normals(O.depth_to_vertices(O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"]), ocam))
        for c in [whole] + slabs:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        whole.sdf_track(k, sp.max_iter_nums, sp.dist_shake, sp.angle_shake)
        for c in slabs:
            assert lib.kf_sdf_partition_begin(c.h, k) == 0
        if k > 0:
            for step in range(sp.max_iter_nums):
                for r, c in enumerate(slabs):
                    assert lib.kf_sdf_partition_step(c.h, step, C.byref(sp), C.byref(c.cam), C.c_void_p(sums[r].data_ptr())) == 0
                    c.sync()
                total = sums[0].clone()
                for s_ in sums[1:]:
                    total += s_                                   # the all-reduce
                for s_ in sums:
                    s_.copy_(total)
                torch.cuda.synchronize()
            for r, c in enumerate(slabs):
                assert lib.kf_sdf_partition_finish(c.h, C.byref(sp), C.byref(c.cam), C.c_void_p(sums[r].data_ptr())) == 0
            ok_o, o_pose, it_o = O.sdf_estimate(ovol, tr, ocam, sp.max_iter_nums, sp.dist_shake, sp.angle_shake, o_pose)
        ok_w, pose_w, st_w, it_w = whole.track_result()
        res_s = [c.track_result() for c in slabs]
        assert ok_w and all(r[0] and r[2] == st_w and r[3] == it_w for r in res_s)
        for r in res_s[1:]:
            assert np.array_equal(r[1].view(np.uint32), res_s[0][1].view(np.uint32))
        assert np.max(np.abs(res_s[0][1] - pose_w)) < 1e-4
        if k > 0:
            assert ok_o and it_o == it_w and np.max(np.abs(res_s[0][1] - o_pose)) < 1e-4
        # keep all volumes identical: fuse with the WHOLE-volume tracker's pose everywhere (the slab poses differ from it in the last bits)
        for c in [whole] + slabs:
            c.integrate(pose_w, trunc, 2.5)
            c.set_pose(pose_w)
        O.integrate(ovol, tr, nn, None, False, False, pose_w, trunc, 2.5, ocam, ocam)
        o_pose = pose_w.copy()
    # a halo thinner than the lookups reach is refused
    thin = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, slab=PL.slab_ranges(res, 2)[1], halo=0)
    assert lib.kf_sdf_partition_step(thin.h, 0, C.byref(sp), C.byref(thin.cam), C.c_void_p(sums[0].data_ptr())) == 1001      # KF_ERR_ARG
    thin.close()
    for c in [whole] + slabs:
        c.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("world,new_cuts", [(2, [0, 40, 128]), (3, [0, 88, 96, 128]), (3, [0, 16, 40, 128])])
def test_layers_change_owner_between_slab_contexts(world, new_cuts):
    """Dynamic slab boundaries on the device side: pipeline.plan_migration's pieces leave their old owner through kf_download_volume_device (pending
    deferred weights applied on the way), every context takes its new range with kf_resize_slab (layers stored before and after keep their voxels),
    the pieces arrive through kf_upload_volume_device -- and the slabs go on fusing.  Before and after, every stored layer (own + halo) equals the
    whole-volume context bit for bit; so do the concatenated marching-cubes extractions and the per-slab update counts' sum."""
    cam, size, res = S.vga_camera(), 3.0, 128
    kcam = K.camera(*cam)
    trunc = 5 * size / res
    halo = PL.slab_halo_layers(res, size, 0.7 * trunc)
    old = PL.slab_ranges(res, world)
    new = [(new_cuts[i], new_cuts[i + 1]) for i in range(world)]
    whole = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=600000)
    slabs = [K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=600000, slab=r, halo=halo) for r in old]
    for c in [whole] + slabs:
        c.set_defer(1)                                            # pending counts exist when the layers leave (frames 1, 2 defer whole free-space quarters)

    def fuse(k):
        pose = S.trajectory_pose(k, size).astype(np.float32)
        mm = S.render_depth_mm(pose, cam, size)
        for c in [whole] + slabs:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.integrate(pose, trunc, 2.5)

    def check():
        for c in slabs:
            z0, z1 = c.stored
            tw, ww = whole.download_volume(z0, z1)
            t, w = c.download_volume(z0, z1)
            assert np.array_equal(bits(t), bits(tw)) and np.array_equal(bits(w), bits(ww)), (c.owned, c.stored)
        assert sum(c.stats()["weight_gt0"] for c in slabs) == whole.stats()["weight_gt0"]

    for k in range(3):
        fuse(k)
    check()
    plan = PL.plan_migration(old, new, halo, res)
    assert plan
    dev = torch.device("cuda", 0)
    pieces = []
    for s, d, z0, z1 in plan:                                      # every piece leaves its old owner before anybody resizes
        t = torch.empty((2, z1 - z0, res, res), dtype=torch.float32, device=dev)
        slabs[s].download_volume_device(z0, z1, t[0].data_ptr(), t[1].data_ptr())
        slabs[s].sync()
        pieces.append(t)
    for c, r in zip(slabs, new):
        c.resize_slab(r[0], r[1], halo)
        assert c.owned == tuple(r) and c.stored == PL.stored_range(r, halo, res)
    for (s, d, z0, z1), t in zip(plan, pieces):
        slabs[d].upload_volume_device(z0, z1, t[0].data_ptr(), t[1].data_ptr())
        slabs[d].sync()
    check()                                                        # nothing lost, nothing invented
    for k in range(3, 6):
        fuse(k)
    check()
    thr = 300 * size / res
    whole.marching_cubes(thr)
    wt = whole.triangles()
    parts = []
    for c in slabs:
        c.marching_cubes(thr)
        parts.append(c.triangles())
    cat = np.concatenate(parts)
    assert len(wt) > 1000 and np.array_equal(cat.view(np.uint32), wt.view(np.uint32))
    for c in [whole] + slabs:
        c.close()


def test_layer_work_counts_the_updates_per_brick_layer():
    """kf_count_layer_work / kf_read_layer_work: the work measure the slab boundaries are balanced on.  With deferral off nothing is retired by the
    cull, so a sampled frame's counts are its update counts per brick layer -- compared with the oracle's (first frame into an empty volume: the
    updated voxels are the ones whose weight became 1); a slab context reports its stored layers only; unsampled frames add nothing."""
    cam, size, res = S.vga_camera(), 3.0, 128
    kcam, ocam = K.camera(*cam), O.Cam.make(*cam)
    trunc = 5 * size / res
    pose = S.trajectory_pose(0, size).astype(np.float32)
    mm = S.render_depth_mm(pose, cam, size)
    tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
    nrm = O.vertices_to_normals(O.depth_to_vertices(O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"]), ocam))
    ovol = O.OVolume(res, size, P["volume_max_weight"])
    n_o = O.integrate(ovol, tr, nrm, None, False, False, pose, trunc, 2.5, ocam, ocam)
    want = (ovol.weight > 0).reshape(res // 8, -1).sum(axis=1).astype(np.uint64)
    assert int(want.sum()) == n_o
    whole = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    slab = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, slab=(48, 88), halo=8)
    for c in (whole, slab):
        c.set_defer(0)
        c.count_layer_work(1)
        for _ in range(2):                                        # the second frame is not sampled
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.integrate(pose, trunc, 2.5)
    got = whole.read_layer_work()
    assert np.array_equal(got, want), (got, want)
    gs = slab.read_layer_work()
    lo, hi = slab.stored[0] // 8, slab.stored[1] // 8
    assert np.array_equal(gs[lo:hi], want[lo:hi]) and not gs[:lo].any() and not gs[hi:].any()
    assert not whole.read_layer_work().any()                       # read with reset: cleared
    whole.close(); slab.close()


def test_vertex_extrapolated_out_of_the_crossing_slab():
    """Found in round 4 (tools/debug_slab_mismatch.py): the reference's vertex is org + dir * alpha with alpha = t - inc * f(t) / (f(t) - f(t - inc))
    (raycastingVolume.cu:89-90) -- an extrapolation whenever both interpolated values have the same sign, e.g. behind an isolated negative voxel at a
    sphere's silhouette -- and then it lands far along the ray, in ANOTHER slab (here: crossing at layer 93, vertex at layer 134.75 on the floor).  The
    merge therefore lets the vertex's owner evaluate the gradient.  A camera that dollies into the scene produces such a pixel at frame 13: every frame's
    merged maps must equal the whole-volume raycast bit for bit, and at least one winner's vertex must be owned by a slab that did not meet its crossing."""
    cam, size, res = S.vga_camera(), 3.0, 256
    kcam = K.camera(*cam)
    inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
    halo = PL.slab_halo_layers(res, size, inc)
    ranges = PL.slab_ranges(res, 2)
    whole = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
    slabs = [K.Context(kcam, res, size, P["volume_max_weight"], levels=3, slab=r, halo=halo) for r in ranges]
    dev = torch.device("cuda", 0)
    for c in [whole] + slabs:
        c.set_pose(S.pose0(size))
    foreign = 0
    for k in range(14):
        p = S.trajectory_pose(k, size)
        p = p @ np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.4 * k / 31.0], [0, 0, 0, 1.0]])
        mm = S.render_depth_mm(p, cam, size)
        for c in [whole] + slabs:
            c.upload_depth_mm(mm)
            c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
            c.integrate(None, P["integrate_sdf_trunc"], P["integrate_depth_trunc"])
        whole.raycast(None, inc, P["depth_trunc_min"], P["depth_trunc_max"])
        wv, wn = whole.download_map(K.MAP_MODEL_VERTICES), whole.download_map(K.MAP_MODEL_NORMALS)
        tas = []
        for c in slabs:
            ta = torch.empty((cam[1], cam[0]), dtype=torch.int64, device=dev)
            c.raycast_slab_cross(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta.data_ptr())
            c.sync()
            tas.append(ta)
        ta_min = torch.stack(tas).min(dim=0).values.contiguous()
        acc = torch.zeros((cam[1], cam[0], 3), dtype=torch.int32, device=dev)
        for r, c in enumerate(slabs):
            cand = _normals_both_forms(c, cam, inc, tas[r], ta_min, dev)
            acc += cand.view(torch.int32)
            foreign += int(((cand.view(torch.int32) != 0).any(dim=-1) & (tas[r] != ta_min)).sum())           # this slab owns the vertex of a crossing another slab met
        rays = acc.view(torch.float32).contiguous()
        for c in slabs:
            c.set_model_maps_rays(None, ta_min.data_ptr(), rays.data_ptr())
            c.sync()
            assert np.array_equal(c.download_map(K.MAP_MODEL_VERTICES).view(np.uint32), wv.view(np.uint32)), k
            assert np.array_equal(c.download_map(K.MAP_MODEL_NORMALS).view(np.uint32), wn.view(np.uint32)), k
    assert foreign >= 1
    for c in [whole] + slabs:
        c.close()
