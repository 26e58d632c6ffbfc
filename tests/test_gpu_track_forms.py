"""GPU: every launch form of the device ICP against the oracle's CameraPoseFinderICP loop (src/CameraPoseFinderICP.cpp:50-145).

kf_icp_track runs either ONE persistent launch (k_icp_loop) or one launch per Gauss-Newton step (k_icp_step + k_track_finish): the
latter whenever a second context lives
on the device, the GPU is shared with another process, or after a time-out.  An image that needs more workgroups than the chip holds at
once (1280x960, BASELINE config C5) takes the BATCHED persistent loop (k_icp_loop_batched).  All forms deal the pixels, reduce and fold in the same
order, so they must agree BITWISE; each is compared with the oracle at the north star's 1e-4 m / 1e-4 rad on identical maps.
The pixel-partitioned form (kf_icp_partition_*, the multi-GPU all-reduce of the 27-float system) is compared with the oracle too.
"""
import ctypes as C

import numpy as np
import pytest
from conftest import default_forms, needs_default_forms

import oracle_lib as O
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import scene as S
from test_gpu_parity import _tracking_case, mid_cam, ragged_cam

pytestmark = pytest.mark.gpu


def _is(form, want):
    """Launch-form check; only binding with the default KF_* form knobs (conftest.FORM_KNOBS) -- results are checked regardless."""
    return form == want or not default_forms()
P = S.STOCK
ICP = (P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])


def _oracle_pose(maps, ocam, pose, levels=3, icp=ICP):
    v, n, ov, on, _ = maps
    return O.icp_estimate(O.pyramid(v, levels), O.pyramid(n, levels, True), O.pyramid(ov, levels), O.pyramid(on, levels, True), ocam, *icp, pose)


def _track(ctx, pose, icp=ICP):
    ctx.set_pose(pose)
    ctx.icp_track(1, *icp)
    ok, p, status, iters = ctx.track_result()
    return ok, p, status, iters, ctx.last_form, ctx.read_solver_params()


@pytest.mark.parametrize("res,cam", [(128, S.vga_camera()), (64, mid_cam()), (96, ragged_cam())])
def test_persistent_loop_and_per_step_launches_agree_bitwise(res, cam):
    size, trunc = 3.0, 5 * 3.0 / res
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(res, size, cam, trunc)
    ok_o, pose_o = _oracle_pose(maps, ocam, pose)
    ok1, p1, st1, it1, form1, sums1 = _track(ctx, pose)
    assert _is(form1, 1)                                        # alone on the device: the persistent loop
    other = K.Context(K.camera(*mid_cam()), 32, 3.0, levels=3)      # a second live context: co-residency is no longer guaranteed
    ok2, p2, st2, it2, form2, sums2 = _track(ctx, pose)
    other.close()
    assert _is(form2, 2)                                        # one launch per step
    assert ok1 and ok2 and ok_o and st1 == st2 == 0 and it1 == it2 == 19
    assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32))            # same pose bits
    assert np.array_equal(sums1.view(np.uint32), sums2.view(np.uint32))      # same final 27 sums
    for p in (p1, p2):
        assert np.max(np.abs(p[:3, 3] - pose_o[:3, 3])) < 1e-4 and np.max(np.abs(p[:3, :3] - pose_o[:3, :3])) < 1e-4
    ok3, p3, _, _, form3, _ = _track(ctx, pose)              # alone again: back on the loop, same bits
    assert _is(form3, 1) and np.array_equal(p3.view(np.uint32), p1.view(np.uint32))
    ctx.close()


@needs_default_forms
def test_forms_at_1280x960_against_the_oracle_and_each_other():
    """BASELINE config C5's image: level 0 needs 800 workgroups, more than the chip holds at once.  Alone on the device the BATCHED persistent loop runs
    (k_icp_loop_batched: ~200 resident workgroups each play several workgroups of the dealing per step); with a second live context one launch per
    step.  Same dealing, reduction and fold -> the same pose bits and final sums; each within 1e-4 of the oracle; the lost verdict through both;
    a time-out of the batched loop is finished by one workgroup, same bits again."""
    cam, res, size = S.vga_camera(2), 128, 3.0
    trunc = 5 * size / res
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(res, size, cam, trunc)
    ok_o, pose_o = _oracle_pose(maps, ocam, pose)
    ok, p, status, iters, form, sums = _track(ctx, pose)
    assert _is(form, 1) and ok and ok_o and status == 0 and iters == 19
    assert np.max(np.abs(p[:3, 3] - pose_o[:3, 3])) < 1e-4 and np.max(np.abs(p[:3, :3] - pose_o[:3, :3])) < 1e-4
    assert np.linalg.norm(p[:3, 3] - nxt[:3, 3]) < np.linalg.norm(pose[:3, 3] - nxt[:3, 3]) + 1e-3
    other = K.Context(K.camera(*mid_cam()), 32, 3.0, levels=3)      # a second live context: one launch per step
    ok2, p2, st2, it2, form2, sums2 = _track(ctx, pose)
    assert _is(form2, 2) and ok2 and st2 == 0 and it2 == 19
    assert np.array_equal(p.view(np.uint32), p2.view(np.uint32)) and np.array_equal(sums.view(np.uint32), sums2.view(np.uint32))
    # lost verdict through both forms: shake threshold 0 rejects the first step, pose unchanged
    ok_l, p_l, st_l, it_l, form_l, _ = _track(ctx, pose, (ICP[0], ICP[1], 0.0, 0.0))
    assert _is(form_l, 2) and not ok_l and st_l == 2 and it_l == 0 and np.array_equal(p_l, pose)
    other.close()
    ok_l, p_l, st_l, it_l, form_l, _ = _track(ctx, pose, (ICP[0], ICP[1], 0.0, 0.0))
    assert _is(form_l, 1) and not ok_l and st_l == 2 and it_l == 0 and np.array_equal(p_l, pose)
    ctx.inject_track_stall(1)                                       # one resident workgroup plays dead: the others time out, one finishes alone
    ok3, p3, st3, it3, form3, sums3 = _track(ctx, pose)
    assert _is(form3, 3) and ok3 and st3 == 0 and it3 == 19
    assert np.array_equal(p.view(np.uint32), p3.view(np.uint32)) and np.array_equal(sums.view(np.uint32), sums3.view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("parts", [2, 3])
def test_pixel_partitioned_icp_against_the_oracle(parts):
    """--icp-mode allreduce: `parts` contexts play the ranks on identical maps, the 27-float systems are added per step (the
    all-reduce) and every rank applies the sum: pose within 1e-4 of the oracle, verdict and iteration count equal, ranks bitwise equal."""
    import torch
    cam, res, size = S.vga_camera(), 128, 3.0
    trunc = 5 * size / res
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(res, size, cam, trunc)
    v, n, ov, on, _ = maps
    ok_o, pose_o = _oracle_pose(maps, ocam, pose)
    ranks = [ctx] + [K.Context(K.camera(*cam), 32, size, levels=3) for _ in range(parts - 1)]
    for c in ranks:
        c.upload_map(K.MAP_NEW_VERTICES, 0, v); c.upload_map(K.MAP_NEW_NORMALS, 0, n)
        c.upload_map(K.MAP_MODEL_VERTICES, 0, ov); c.upload_map(K.MAP_MODEL_NORMALS, 0, on)
        c.set_pose(pose)
    lib = K.load()
    icp = K.IcpParams(3, P["icp_thre_sin_angle"], P["icp_thre_dist"], P["camera_shake_dist"], P["camera_shake_angle"])
    sums = [torch.zeros(32, dtype=torch.float32, device="cuda") for _ in ranks]
    for c in ranks:
        assert lib.kf_icp_partition_begin(c.h, 1) == 0
    for step in range(lib.kf_icp_partition_steps(ctx.h)):
        for r, c in enumerate(ranks):
            assert lib.kf_icp_partition_step(c.h, step, C.byref(icp), C.byref(c.cam), r, parts, C.c_void_p(sums[r].data_ptr())) == 0
            c.sync()
        total = sums[0].clone()
        for s_ in sums[1:]:
            total += s_                                       # rank order, as a ring all-reduce of three would fix it
        for s_ in sums:
            s_.copy_(total)
        torch.cuda.synchronize()
    for r, c in enumerate(ranks):
        assert lib.kf_icp_partition_finish(c.h, C.byref(icp), C.c_void_p(sums[r].data_ptr())) == 0
    res_r = [c.track_result() for c in ranks]
    assert ok_o and all(r[0] and r[2] == 0 and r[3] == 19 for r in res_r)
    for r in res_r[1:]:
        assert np.array_equal(r[1].view(np.uint32), res_r[0][1].view(np.uint32))
    p = res_r[0][1]
    assert np.max(np.abs(p[:3, 3] - pose_o[:3, 3])) < 1e-4 and np.max(np.abs(p[:3, :3] - pose_o[:3, :3])) < 1e-4
    for c in ranks:
        c.close()


@pytest.mark.parametrize("icp", [(-0.1, ICP[1], ICP[2], ICP[3]), (ICP[0], -0.1, ICP[2], ICP[3]), (ICP[0], ICP[1], -0.3, ICP[3]), (ICP[0], ICP[1], ICP[2], -0.3)])
def test_negative_thresholds_reject_like_the_reference(icp):
    """`norm > negative` holds for every pixel / every increment in the reference (CalPointToPlaneErrSolverParams.cu:52,
    CameraPoseFinderICP.cpp:101-107): no correspondence survives (singular system) resp. every step counts as camera shake."""
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(64, 3.0, mid_cam(), 5 * 3.0 / 64)
    ok_o, _ = _oracle_pose(maps, ocam, pose, icp=icp)
    ok, p, status, iters, form, _ = _track(ctx, pose, icp)
    assert not ok_o and not ok and status in (1, 2) and np.array_equal(p, pose)
    assert status == (1 if (icp[0] < 0 or icp[1] < 0) else 2)
    ctx.close()


@needs_default_forms
@pytest.mark.parametrize("res,cam", [(128, S.vga_camera()), (96, ragged_cam())])
def test_a_loop_that_times_out_is_finished_by_one_workgroup_with_the_same_bits(res, cam):
    """kf_inject_track_stall: one workgroup of the persistent loop plays dead (as if a foreign process had kept it off the chip).  The others
    time out, one of them claims the launch and runs the frame's whole Gauss-Newton loop alone, playing every workgroup in turn: the same pose
    bits and final 27 sums as the undisturbed loop, launch_form 3, tracked -- the frame is not lost; the context then backs off to per-step
    launches (form 2) for a while.  Also the lost verdict through the solo path."""
    size, trunc = 3.0, 5 * 3.0 / res
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(res, size, cam, trunc)
    ok1, p1, st1, it1, form1, sums1 = _track(ctx, pose)
    assert _is(form1, 1) and ok1 and st1 == 0 and it1 == 19
    ctx.inject_track_stall(1)
    ok2, p2, st2, it2, form2, sums2 = _track(ctx, pose)
    assert _is(form2, 3), form2                                  # timed out, finished solo
    assert ok2 and st2 == 0 and it2 == 19
    assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)) and np.array_equal(sums1.view(np.uint32), sums2.view(np.uint32))
    ok3, p3, st3, it3, form3, _ = _track(ctx, pose)          # back-off: one launch per step now, same bits again
    assert _is(form3, 2) and np.array_equal(p3.view(np.uint32), p1.view(np.uint32))
    ctx.close()
    # the lost verdict on the solo path: shake threshold 0 rejects the first step, pose unchanged, iterations 0 (not a stale count)
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(res, size, cam, trunc)
    _track(ctx, pose)
    ctx.inject_track_stall(1)
    ok_l, p_l, st_l, it_l, form_l, _ = _track(ctx, pose, (ICP[0], ICP[1], 0.0, 0.0))
    assert not ok_l and st_l == 2 and it_l == 0 and np.array_equal(p_l, pose)
    ctx.close()


@needs_default_forms
def test_a_timed_out_loop_in_the_streamed_pipeline_loses_no_frame():
    """The asynchronous pipeline (no host synchronisation per frame, next frame's front end riding in the launches): a loop launch that times out
    in the middle of the stream costs milliseconds, not the frame -- frames_lost stays 0 and every pose equals the undisturbed run's bit for bit."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    cam, res, size = S.vga_camera(), 384, 3.0                    # stock truncation (0.05 m) = 6.4 voxels, as at C2
    n = 8
    frames, _ = S.make_stream(n, cam, size)
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    poses, forms, vols = [], [], []
    for inject_at in (None, 4):
        pipe = SingleGpuPipeline(K.camera(*cam), res, size, dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"]))
        out, fo = [], []
        for k in range(n):
            if inject_at == k:
                pipe.ctx.inject_track_stall(1)
            pipe.process_frame_device(dev.data_ptr() + k * fb, k, dev.data_ptr() + ((k + 1) % n) * fb)
            ok, pose, status, iters = pipe.track_result()
            assert ok and status == 0, (inject_at, k)
            out.append(pose); fo.append(pipe.ctx.last_form)
        st = pipe.stats()
        assert st["frames_lost"] == 0 and st["frames_fused"] == n
        # (the fusion pass's cull ran as the tail of the loop launches -- also of the one that was finished by a single workgroup)
        consumed, undone = pipe.ctx.cull_tail_counts()
        assert undone == 0 and consumed >= (n - 1 if inject_at is None else 2), (consumed, undone)
        poses.append(out); forms.append(fo); vols.append((pipe.ctx.download_volume(), st["updated_total"]))
        pipe.close()
    assert forms[1][4] == 3 and forms[1][5] == 2 and forms[0][4] == 1
    for a, b in zip(*poses):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    (va, ua), (vb, ub) = vols
    assert ua == ub and np.array_equal(va[0].view(np.uint32), vb[0].view(np.uint32)) and np.array_equal(va[1], vb[1])


@needs_default_forms
def test_the_cull_in_the_tracking_launch_equals_the_cull_launch():
    """Three launches per streamed frame: the persistent loop's workgroups run the fusion pass's brick cull as their tail once the pose is committed
    (cull.h), kf_integrate_volume consumes it.  Three runs of one stream fuse the same voxels frame by frame (poses, queue lengths, update counts,
    voxel bits): every tail consumed; tails undone along the way (the caller supplies the pose itself, sets the pose in between, asks for another
    integration distance); no tail at all (a second live context: one launch per Gauss-Newton step, cull in a launch of its own)."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    cam, res, size = S.vga_camera(), 256, 3.0
    n = 9
    frames, _ = S.make_stream(n, cam, size)
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    wl = dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"])
    runs = {}
    for mode in ("tails", "mixed", "no-tails"):
        other = K.Context(K.camera(64, 48, 31.5, 23.5, 52.5, 52.5), 32, 3.0, levels=3) if mode == "no-tails" else None
        pipe = SingleGpuPipeline(K.camera(*cam), res, size, wl)
        c = pipe.ctx
        log = []
        for k in range(n):
            c.set_depth_mm_device(dev.data_ptr() + k * fb)
            c.preprocess(P["depth_trunc_min"], pipe.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
            if k + 1 < n:
                c.prefetch_frame(dev.data_ptr() + (k + 1) * fb, P["depth_trunc_min"], pipe.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
            c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
            ok, pose, status, iters = pipe.track_result()
            assert ok
            dist = pipe.integ_dist * (0.5 if (k == n - 1 and mode != "tails") else 1.0)   # the last frame of two runs: not what the tail culled for
            if mode == "mixed" and k == 2:
                c.integrate(pose, P["integrate_sdf_trunc"], dist)                        # the caller's own copy of the pose: the tail is undone, the cull runs again
            elif mode == "mixed" and k == 4:
                c.set_pose(pose)                                                          # (the same pose: the tail is void all the same)
                c.integrate(None, P["integrate_sdf_trunc"], dist)
            else:
                c.integrate(None, P["integrate_sdf_trunc"], dist)
            st = c.stats()
            log.append((pose.copy(), st["bricks_active"], st["updated_last"]))
            c.raycast(None, pipe.inc, P["depth_trunc_min"], pipe.trunc_max)
        pipe.sync()
        runs[mode] = (log, c.download_volume(), c.stats()["updated_total"], c.cull_tail_counts(), c.stats()["frames_fused"])
        pipe.close()
        if other is not None:
            other.close()
    assert runs["tails"][3] == (n - 1, 0)                         # (frame 0 has no tracking launch; its fusion pass leaves the first hint)
    assert runs["mixed"][3] == (n - 1 - 4, 3)                     # (three undone; and no tail in the frame after the caller's own pose: no hint)
    assert runs["no-tails"][3] == (0, 0)
    for mode in ("mixed", "no-tails"):
        last = n if mode == "no-tails" else n - 1                  # "tails" fused its last frame with the full distance
        for k in range(last):
            ra, rb = runs["mixed" if mode == "no-tails" else "tails"][0][k], runs[mode][0][k]
            assert np.array_equal(ra[0].view(np.uint32), rb[0].view(np.uint32)) and ra[1:] == rb[1:], (mode, k, ra[1:], rb[1:])
    (_, va, ua, _, fa), (_, vb, ub, _, fb_) = runs["mixed"], runs["no-tails"]
    assert ua == ub and fa == fb_ == n and np.array_equal(va[0].view(np.uint32), vb[0].view(np.uint32)) and np.array_equal(va[1], vb[1])
    assert runs["tails"][0][5][2] > 100000 and runs["mixed"][0][n - 1][2] < runs["tails"][0][n - 1][2]


# ---- the SDF tracker's launch forms (CameraPoseFinderSDF, src/CameraPoseFinderSDF.cpp:45-105) ---------------------------------------------
SDF = (P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])


def _track_sdf(ctx, pose, sdf=SDF):
    ctx.set_pose(pose)
    ctx.sdf_track(1, *sdf)
    ok, p, status, iters = ctx.track_result()
    return ok, p, status, iters, ctx.last_form, ctx.read_solver_params()


@pytest.mark.parametrize("res,cam", [(128, S.vga_camera()), (64, mid_cam()), (96, ragged_cam())])
def test_sdf_persistent_loop_against_the_oracle_and_the_per_iteration_form(res, cam):
    """kf_sdf_track alone on the device runs ONE launch for the whole Gauss-Newton loop (k_sdf_loop: tagged partial sums, wave solve, exit on convergence);
    with a second live context one launch per iteration (k_sdf_step).  The two deal pixels to workgroups differently (fp32 sums in another order), so
    they agree to tolerance, each within 1e-4 of the oracle's CameraPoseFinderSDF loop, with the same iteration count and verdict."""
    size, trunc = 3.0, 5 * 3.0 / res
    ctx, ovol, pose, nxt, ocam, (v, n, ov, on, tr) = _tracking_case(res, size, cam, trunc, n_warm=3)
    ok_o, pose_o, it_o = O.sdf_estimate(ovol, tr, ocam, *SDF, pose)
    ok1, p1, st1, it1, form1, sums1 = _track_sdf(ctx, pose)
    assert _is(form1, 1)
    other = K.Context(K.camera(*mid_cam()), 32, 3.0, levels=3)
    ok2, p2, st2, it2, form2, sums2 = _track_sdf(ctx, pose)
    other.close()
    assert _is(form2, 2)
    assert ok_o and ok1 and ok2 and st1 == st2 == 0 and it1 == it2 == it_o, (it1, it2, it_o)
    for p in (p1, p2):
        assert np.max(np.abs(p - pose_o)) < 1e-4
    assert np.max(np.abs(sums1 - sums2)) <= 2e-5 * np.max(np.abs(sums2))     # the last iteration's system, two summation orders
    ok3, p3, _, it3, form3, sums3 = _track_sdf(ctx, pose)                      # the loop again: reproducible to the bit
    assert _is(form3, 1) and np.array_equal(p3.view(np.uint32), p1.view(np.uint32)) and np.array_equal(sums3.view(np.uint32), sums1.view(np.uint32))
    # the lost verdict: a shake threshold of 0 rejects the first increment, the pose stays, no iteration counts (SDF.cpp:81-85)
    for second in (False, True):
        other = K.Context(K.camera(*mid_cam()), 32, 3.0, levels=3) if second else None
        ok_l, p_l, st_l, it_l, form_l, _ = _track_sdf(ctx, pose, (SDF[0], 0.0, 0.0))
        assert _is(form_l, 2 if second else 1) and not ok_l and st_l == 2 and it_l == 0 and np.array_equal(p_l, pose)
        if other is not None:
            other.close()
    ctx.close()


def test_sdf_loop_at_1280x960_and_with_a_single_iteration():
    """BASELINE config C5's image through the SDF loop (every lane walks ten pixels per iteration), and max_iter_nums = 1 (one pixel phase, one update)."""
    cam, res, size = S.vga_camera(2), 128, 3.0
    trunc = 5 * size / res
    ctx, ovol, pose, nxt, ocam, (v, n, ov, on, tr) = _tracking_case(res, size, cam, trunc, n_warm=3)
    for sdf in (SDF, (1, SDF[1], SDF[2])):
        ok_o, pose_o, it_o = O.sdf_estimate(ovol, tr, ocam, *sdf, pose)
        ok, p, status, iters, form, _ = _track_sdf(ctx, pose, sdf)
        assert _is(form, 1) and ok and ok_o and status == 0 and iters == it_o
        assert np.max(np.abs(p - pose_o)) < 1e-4
    ctx.close()


@needs_default_forms
@pytest.mark.parametrize("res,cam", [(128, S.vga_camera()), (96, ragged_cam())])
def test_an_sdf_loop_that_times_out_is_finished_by_one_workgroup_with_the_same_bits(res, cam):
    """kf_inject_track_stall on the SDF loop: one workgroup plays dead, the others time out, exactly one claims the launch (KfTrackState::commit_word) and
    runs the loop again alone, playing every workgroup of the dealing in turn: the same pose bits and final sums as the undisturbed loop,
    launch_form 3, the frame is kept; then the context backs off to one launch per iteration."""
    size, trunc = 3.0, 5 * 3.0 / res
    ctx, ovol, pose, nxt, ocam, maps = _tracking_case(res, size, cam, trunc, n_warm=3)
    ok1, p1, st1, it1, form1, sums1 = _track_sdf(ctx, pose)
    assert _is(form1, 1) and ok1 and st1 == 0
    ctx.inject_track_stall(1)
    ok2, p2, st2, it2, form2, sums2 = _track_sdf(ctx, pose)
    assert _is(form2, 3), form2
    assert ok2 and st2 == 0 and it2 == it1
    assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)) and np.array_equal(sums1.view(np.uint32), sums2.view(np.uint32))
    ok3, p3, st3, it3, form3, _ = _track_sdf(ctx, pose)
    assert _is(form3, 2) and ok3 and it3 == it1 and np.max(np.abs(p3 - p1)) < 1e-5
    ctx.close()
