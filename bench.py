#!/usr/bin/env python3
"""bench.py -- depth frames/sec through the per-frame KinectFusion path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one depth frame through  u16-mm -> f32 -> gate -> bilateral -> vertices/normals -> 3-level ICP (10/5/4,
device-resident Gauss-Newton) -> TSDF integrate -> raycast.  Frames are synthetic (Scene S, hybkinectfu_amd/scene.py) and
already resident in HBM when the timed region starts.  N=1 runs BASELINE.json configs[1] ("C2": 512^3 @ 4 m, VGA);
N>1 runs configs[3] ("C4": 1024^3 @ 6 m, z-slab per GPU, strong scaling) under torch.distributed/RCCL.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (the TSDF fusion kernel, HBM-bound,
timed with HIP events on the context's own stream) and `cpu_baseline` (the CPU oracle on a bounded sample, N=1 only).
The N=1 line also carries `multi_gpu_workload_on_1_gpu` (C4 unpartitioned on this one GPU, the same-workload reference for the
N=2/4/8 lines), `other_configs` (C1: 256^3 @ 3 m, C3: 512^3 with the SDF tracker, C5: 2048^3 @ 8 m with 1280x960 depth and mesh
extraction, a few dozen frames each) and `steady_state`; the headline `value` at N=1 is C2, as BASELINE.json's metric states.
The N>1 line carries `per_rank` (what every z-slab rank fused and how long its integrate / raycast / merge stages took) and a short
`c5` block (2048^3 @ 8 m on the same ranks).  `config.env` lists every KF_* variable that was set.
`roofline` is the PLAIN read-modify-write form of the fusion kernel (algorithmic bytes / its live dispatch time, measured right after the timed
region with kf_set_defer(0)); `roofline.in_timed_region` is the DEFER form the timed frames run, with the bytes it really moves.  `value` =
`value_resident`; `value_pcie_inclusive` beside it; `parity_witness` compares the HIP path's pose and update count with the oracle's on the
frames the cpu_baseline leg ran; `scene_noise` repeats the headline on a stream with sensor noise.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hybkinectfu_amd import scene as S  # noqa: E402   (numpy only: no GPU, no library load)

P = S.STOCK
K = None                       # hybkinectfu_amd.lib, bound in main() of a RANK process (the launching parent never loads it)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def workload(n_gpus, name="auto"):
    if name == "auto":
        name = "c2" if n_gpus == 1 else "c4"
    if name == "c2":
        return dict(name="C2", res=512, size=4.0, cam=S.vga_camera(), trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"],
                    desc="C2: synthetic 640x480 depth stream (Scene S), 512^3 @ 4 m TSDF, 3-level ICP 10/5/4, %dxMI355X" % n_gpus)
    if name == "c1":
        # C1: BASELINE.json configs[0] is TUM freiburg1_xyz through DataSourceProducerRGBDDataset; the dataset is not in this image, so the
        # same geometry (VGA, 256^3 @ 3 m, 3-level ICP) runs on Scene S (the reader itself is covered by tests/test_gpu_configs.py::C1)
        return dict(name="C1", res=256, size=3.0, cam=S.vga_camera(), trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"],
                    desc="C1 geometry: 640x480 depth (Scene S; the TUM freiburg1_xyz files are not in this image), 256^3 @ 3 m TSDF, 3-level ICP, %dxMI355X" % n_gpus)
    if name == "c3":
        return dict(name="C3", res=512, size=4.0, cam=S.vga_camera(), trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"], tracker="sdf",
                    desc="C3 geometry: 640x480 depth (Scene S; the TUM freiburg3_long_office files are not in this image), 512^3 @ 4 m TSDF, "
                         "CameraPoseFinderSDF (direct SDF tracking, max %d iterations), %dxMI355X" % (int(P["sdf_max_iter_nums"]), n_gpus))
    if name == "c5":
        # C5: 1280x960 depth, 2048^3 @ 8 m (68.7 GB of voxels: one GPU holds it, N GPUs hold a z-slab each), mesh extraction at the end
        return dict(name="C5", res=2048, size=8.0, cam=S.vga_camera(2), trunc_max=8.0, integ_dist=8.0, extract_mesh=True,
                    desc="C5: synthetic 1280x960 depth (Scene S), 2048^3 @ 8 m TSDF + marching-cubes extract, " +
                         ("whole volume on 1 GPU" if n_gpus == 1 else "z-slab per GPU, %d GPUs" % n_gpus))
    # C4: depth gates raised to the volume size so the whole 6 m volume is exercised
    return dict(name="C4", res=1024, size=6.0, cam=S.vga_camera(), trunc_max=6.0, integ_dist=6.0,
                desc="C4: synthetic 640x480 depth (Scene S), 1024^3 @ 6 m TSDF, " +
                     ("whole volume on 1 GPU" if n_gpus == 1 else "z-slab per GPU, %d GPUs" % n_gpus))


def cpu_baseline(wl, frames_mm, n_sample=150, witness=None):
    """The CPU oracle (oracle/, 'port') on the first n_sample frames of the same stream (about 10-15 s of CPU work on the
    GPU box's 16-core share), OpenMP over the box's cores.
    witness = (GPU pose after frame n_sample - 1 of the same frames, voxels the GPU fused on that frame): the oracle's own pose at that
    frame is compared with the GPU's, and the oracle counts the voxels that frame updates under the GPU's pose (the update predicate,
    integrateVolume.cu:39-67, depends on the depth map and the pose only) -- returned as the second value, outside the timed work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    # the GPU box gives one GPU job a 16-core share of the host; more OpenMP threads than that only oversubscribe it
    # (tools/cpu_scaling.py: integrate is fastest at 16 threads there)
    cores = O.set_threads(min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1))
    cam = wl["cam"]
    ocam = O.Cam.make(*cam)
    vol = O.OVolume(wl["res"], wl["size"], P["volume_max_weight"])
    pose = S.pose0(wl["size"])
    mv = mn = None
    ok = True
    t0 = time.perf_counter()
    for k in range(n_sample):
        tr = O.trunc_depth(O.depth_mm_to_m(frames_mm[k % len(frames_mm)]), P["depth_trunc_min"], wl["trunc_max"])
        fl = O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        v = O.depth_to_vertices(fl, ocam)
        n = O.vertices_to_normals(v)
        if k > 0 and wl.get("tracker") == "sdf":
            ok, pose, _ = O.sdf_estimate(vol, tr, ocam, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
        elif k > 0:
            ok, pose = O.icp_estimate(O.pyramid(v, 3), O.pyramid(n, 3, True), O.pyramid(mv, 3), O.pyramid(mn, 3, True), ocam,
                                      P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"], pose)
        O.integrate(vol, tr, n, None, False, False, pose, P["integrate_sdf_trunc"], wl["integ_dist"], ocam, ocam)
        mv, mn, _ = O.raycast(vol, False, pose, P["raycast_increment_factor"] * P["integrate_sdf_trunc"], ocam, P["depth_trunc_min"], wl["trunc_max"])
    dt = time.perf_counter() - t0
    base = dict(value=round(n_sample / dt, 4), unit="frames/s", cores=int(cores), kind="port",
                sample="%d frames of the same stream through oracle/libkforacle.so (preprocess+ICP+integrate+raycast), %.1f s" % (n_sample, dt))
    wit = None
    if witness is not None:
        gpu_pose, gpu_n_upd, gpu_tracked = witness
        from hybkinectfu_amd import posemath as PM
        gp, op = np.asarray(gpu_pose, np.float64).reshape(4, 4), np.asarray(pose, np.float64).reshape(4, 4)
        # angle from the skew part of Rg^T Ro after projecting both blocks onto SO(3) (posemath.rotation_angle): the composed fp32 poses sit
        # ~1e-6 off orthonormal, which arccos of the trace reads as ~2e-3 "rad" (VERDICT r4 / BENCH_r04's false `false`)
        d_t, ang, d_el = PM.pose_difference(gp, op)
        n_o = O.integrate(vol, tr, n, None, False, False, np.asarray(gpu_pose, np.float32), P["integrate_sdf_trunc"], wl["integ_dist"], ocam, ocam)
        wit = dict(frame=n_sample - 1, frames_tracked_by_both=n_sample - 1, oracle_tracked=bool(ok), gpu_tracked=bool(gpu_tracked),
                   pose_translation_diff_m=d_t, pose_rotation_diff_rad=ang, pose_rotation_max_element_diff=d_el,
                   rotation_metric="atan2(|vee(D - D^T)|/2, (tr D - 1)/2), D = Rg^T Ro, both blocks projected to SO(3) (hybkinectfu_amd/posemath.py)",
                   orthonormality_defect_gpu=PM.orthonormality_defect(gp[:3, :3]), orthonormality_defect_oracle=PM.orthonormality_defect(op[:3, :3]),
                   tolerance="1e-4 m / 1e-4 rad (BASELINE.json north_star)",
                   pose_within_tolerance=bool(ok and gpu_tracked and d_t <= 1e-4 and ang <= 1e-4 and d_el <= 1e-4),
                   n_upd_gpu=int(gpu_n_upd), n_upd_oracle_with_gpu_pose=int(n_o), n_upd_equal=bool(int(gpu_n_upd) == int(n_o)),
                   note="the HIP path and the oracle each tracked and fused the same %d frames on their own (the cpu_baseline leg); poses compared at the "
                        "last frame, and the oracle's count of voxels passing the update predicate for that frame under the GPU's pose against the GPU's "
                        "device counter (bit-exact gate; the full parity suite is tests/ -m gpu)" % n_sample)
    return base, wit


def gpu_witness_run(wl, frames_mm, n_sample):
    """the same first n_sample frames through a fresh HIP pipeline: (pose after the last frame, voxels fused on it, tracked)"""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    cam = wl["cam"]
    n_u = len(frames_mm)
    dev = torch.from_numpy(np.ascontiguousarray(frames_mm).astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=torch.cuda.current_device())
    for k in range(n_sample):
        pipe.process_frame_device(dev.data_ptr() + (k % n_u) * fb, k, dev.data_ptr() + ((k + 1) % n_u) * fb)
    pipe.sync()
    tracked, pose, status, iters = pipe.track_result()
    n_last = pipe.stats()["updated_last"]
    pipe.close()
    return pose, n_last, tracked


def env_knobs():
    """Every KF_* variable set in this process's environment (they select library variants and tuning / diagnostic paths: a line
    measured with one of them set says so)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("KF_")}


def ping_pong(k, n):
    """Index into n unique frames of the stream for frame k: 0 .. n-1, n-1 .. 0, ... -- a short cached stretch of the trajectory walked
    back and forth, so consecutive frames stay neighbours of the camera path (a wrap-around would jump)."""
    if n <= 1:
        return 0
    m = k % (2 * n - 2)
    return m if m < n else 2 * n - 2 - m


def single_gpu_reference(name, n_frames=50, warmup=10, n_unique=None, overrides=None):
    """Frames/s of workload `name` on ONE GPU without partitioning: the same-workload reference point of the multi-GPU series
    (BASELINE.json's metric quotes 1024^3 for 1/2/4/8 GPUs, while the N=1 headline line is the 512^3 configuration), and the
    other BASELINE configurations next to the headline (`other_configs`)."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    wl = dict(workload(1, name))
    wl.update(overrides or {})
    cam = wl["cam"]
    n_unique = min(n_frames, 100) if n_unique is None else n_unique
    frames, _ = S.make_stream(n_unique, cam, wl["size"])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    at = (lambda k: k % n_unique) if n_unique >= min(n_frames, 100) else (lambda k: ping_pong(k, n_unique))
    pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=torch.cuda.current_device(),
                             max_triangles=(16_000_000 if wl.get("extract_mesh") else 0))
    if wl.get("color"):
        pipe.ctx.upload_rgb(np.random.default_rng(1).integers(0, 256, (cam[1], cam[0], 3)).astype(np.uint8))
    for k in range(warmup):
        pipe.process_frame_device(dev.data_ptr() + at(k) * fb, k, dev.data_ptr() + at(k + 1) * fb)
    pipe.sync(); torch.cuda.synchronize()
    s0 = pipe.stats()
    pipe.stage_timers((1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (4 << 8))
    t0 = time.perf_counter()
    for k in range(warmup, n_frames):
        pipe.process_frame_device(dev.data_ptr() + at(k) * fb, k, dev.data_ptr() + at(k + 1) * fb)
    pipe.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, cnt = pipe.read_stage_ms()
    s1 = pipe.stats()
    lost = s1["frames_lost"] - s0["frames_lost"]
    out = dict(workload=wl["desc"], value=round((n_frames - warmup) / dt, 2), unit="frames/s", steps=n_frames - warmup,
               ms_per_step=round(1000.0 * dt / (n_frames - warmup), 4), frames_lost=int(lost),
               n_upd_per_frame=int((s1["updated_total"] - s0["updated_total"]) / max(n_frames - warmup, 1)),
               stage_us={STAGE_NAMES[i]: round(1000.0 * float(ms[i]) / max(int(cnt[i]), 1), 2) for i in (2, 3, 4, 5)})
    if wl["res"] >= 768 and not wl.get("color"):
        # the fusion kernel's PLAIN read-modify-write form at this size against the HBM roofline (the timed frames above ran the DEFER form): as the headline's
        # `roofline` leg -- kf_set_defer(0), two frames for the flush, then the kernel's own dispatch time (HIP events) and N_upd over 8 frames
        k0 = n_frames
        pipe.ctx.set_defer(0)
        for k in range(k0, k0 + 2):
            pipe.process_frame_device(dev.data_ptr() + at(k) * fb, k, dev.data_ptr() + at(k + 1) * fb)
        pipe.sync()
        sp0 = pipe.stats()
        pipe.stage_timers((1 << 8) | (1 << 5))
        for k in range(k0 + 2, k0 + 10):
            pipe.process_frame_device(dev.data_ptr() + at(k) * fb, k, dev.data_ptr() + at(k + 1) * fb)
        pipe.sync()
        pms, pcnt = pipe.read_stage_ms()
        sp1 = pipe.stats()
        pipe.stage_timers(0)
        pipe.ctx.set_defer(-1)
        if int(pcnt[5]) >= 5 and float(pms[5]) > 0:
            k_ms = float(pms[5]) / int(pcnt[5])
            alg = (sp1["updated_total"] - sp0["updated_total"]) / 8.0 * 16.0 + cam[0] * cam[1] * 4.0
            out["plain_kernel_roofline"] = dict(bound="hbm", kernel="k_integrate_pairs (plain read-modify-write form: deferral off)", kernel_ms=round(k_ms, 5), launches_timed=int(pcnt[5]),
                                                algorithmic_bytes_per_launch=int(alg), achieved=round(alg / (k_ms * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                                frac=round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=profiled_traffic(wl["name"])[0],
                                                measured="live, HIP events stamped by the kernel's own dispatch, 8 frames behind this block's timed frames with kf_set_defer(0)")
    if wl.get("extract_mesh"):
        import ctypes as C
        pipe.ctx.marching_cubes(300.0 * wl["size"] / wl["res"])        # first call allocates the extraction's scratch
        pipe.ctx.clear_triangles(); pipe.sync()
        t0m = time.perf_counter()
        pipe.ctx.marching_cubes(300.0 * wl["size"] / wl["res"])
        n_tri = C.c_uint32()
        K._chk(pipe.ctx.lib.kf_triangle_count(pipe.ctx.h, C.byref(n_tri)), "kf_triangle_count")
        out["mesh_extraction"] = dict(triangles=int(n_tri.value), ms=round(1000.0 * (time.perf_counter() - t0m), 3),
                                      note="second kf_marching_cubes call (the first allocates its scratch), wall time incl. the count read-back")
    pipe.close()
    return out


def steady_state(name, n_timed=100):
    """The same stream after max_weight + 32 frames: free space the camera keeps looking through has saturated at (tsdf 1, weight
    max_weight); ANY free-space observation of a saturated quarter brick is the identity (before saturation only whole-quarter
    observations are deferred: integrate.hip, k_integrate_pairs<.., DEFER>).  Reported next to the headline: frames/s, the fusion
    kernel's time, the reference's bytes (N_upd x 16 B + the depth image) over that time -- not an HBM rate -- and what the kernel moves."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    wl = workload(1, name)
    cam = wl["cam"]
    frames, _ = S.make_stream(100, cam, wl["size"])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=torch.cuda.current_device())
    n_pre = int(P["volume_max_weight"]) + 32
    for k in range(n_pre):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    pipe.sync(); torch.cuda.synchronize()
    s0 = pipe.stats()
    pipe.stage_timers((1 << 5) | (1 << 3) | (4 << 8))
    t0 = time.perf_counter()
    for k in range(n_pre, n_pre + n_timed):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    pipe.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, cnt = pipe.read_stage_ms()
    s1 = pipe.stats()
    pipe.close()
    n_upd = (s1["updated_total"] - s0["updated_total"]) / n_timed
    alg = n_upd * 16.0 + cam[0] * cam[1] * 4.0
    k_ms = float(ms[5]) / max(int(cnt[5]), 1)
    out = dict(workload=wl["desc"], frames_fused_before=n_pre, steps=n_timed, value=round(n_timed / dt, 2), unit="frames/s",
               frames_lost=int(s1["frames_lost"] - s0["frames_lost"]), kernel=("k_integrate_pairs<.., DEFER> past saturation" if wl["res"] >= 768 else "k_integrate_pairs (plain: volumes below 768^3 do not defer by default)"), kernel_ms=round(k_ms, 5),
               launches_timed=int(cnt[5]), reference_bytes_per_launch=int(alg),
               reference_bytes_rate=round(alg / (k_ms * 1e-3) / 1e9, 2) if k_ms > 0 else None, unit_rate="GB/s",
               reference_bytes_rate_over_peak=round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
               note="reference_bytes = what the REFERENCE's update moves for these frames (N_upd x 16 B + the depth image); the kernel itself moves fewer -- "
                    "deferred / saturated free space is counted, not touched -- so reference_bytes_rate is NOT an HBM rate and may exceed the 8 TB/s peak; "
                    "kernel_traffic_* is what the kernel really moves")
    # what the SAT kernel really moves per launch: rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE, separate passes) from the builder's own profile run
    tpath = os.path.join(ROOT, "profiles", "integrate_traffic.json")
    try:
        tj = json.load(open(tpath))
        tr = tj.get(wl["name"] + "_saturated") if wl["res"] >= 768 else tj.get(wl["name"])      # (below 768^3 the plain kernel runs: its own traffic figure)
    except Exception:
        tj, tr = {}, None
    out["kernel_traffic_bytes_per_launch"] = tr
    out["kernel_traffic_rate"] = round(tr / (k_ms * 1e-3) / 1e9, 2) if (tr and k_ms > 0) else None
    out["kernel_traffic_frac_of_hbm_peak"] = round(tr / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (tr and k_ms > 0) else None
    out["kernel_traffic_source"] = tj.get("source") if tr else None
    return out


def scene_noise_block(wl, n_frames=110, warmup=10):
    """The headline workload on a NOISY stream: Scene S with the LCG-12345 sensor noise of scene.add_sensor_noise (axial noise growing with depth,
    2 % drop-outs).  Every other number of the line comes from the noise-free scene, and several tuning constants were fitted to it; this block
    says what the same pipeline does when bilateral early returns, ICP rejections, ray stragglers and partial fusion waves look like a sensor's.
    Parity on these very frames: tests/test_gpu_parity.py::test_noisy_scene_parity."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    cam = wl["cam"]
    frames, poses = S.make_stream(100, cam, wl["size"])
    frames = S.add_sensor_noise(frames)
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=torch.cuda.current_device())
    for k in range(warmup):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    pipe.sync(); torch.cuda.synchronize()
    s0 = pipe.stats()
    t0 = time.perf_counter()
    for k in range(warmup, n_frames):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    pipe.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = pipe.stats()
    tracked, pose, status, iters = pipe.track_result()
    gt = poses[(n_frames - 1) % 100]
    pipe.stage_timers(0x1F | (1 << 5))
    for k in range(n_frames, n_frames + 40):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
    pipe.sync()
    sm, sc = pipe.read_stage_ms()
    pipe.close()
    return dict(workload=wl["desc"] + " + sensor noise (LCG seed 12345: sigma(z) = 1.2 mm + 1.9 mm/m^2 (z - 0.4 m)^2, 2 % drop-outs)",
                value=round((n_frames - warmup) / dt, 2), unit="frames/s", steps=n_frames - warmup, ms_per_step=round(1000.0 * dt / (n_frames - warmup), 4),
                frames_lost=int(s1["frames_lost"] - s0["frames_lost"]), tracked_last_frame=bool(tracked), iterations_last_frame=int(iters),
                n_upd_per_frame=int((s1["updated_total"] - s0["updated_total"]) / max(n_frames - warmup, 1)),
                distance_to_ground_truth_m=float(np.linalg.norm(np.asarray(pose, np.float64)[:3, 3] - gt[:3, 3])),
                stage_us={STAGE_NAMES[i]: round(1000.0 * float(sm[i]) / max(int(sc[i]), 1), 2) for i in (1, 2, 3, 4, 5)})


def stage_block(alg_bytes, stage_ms, how):
    return dict(kernels="whole integrate stage: k_integrate_pairs, plus k_integrate_cull where it is a launch of its own (at C2 the cull runs as the tail of the tracking launch)", ms=round(stage_ms, 5), measured=how,
                achieved=round(alg_bytes / (stage_ms * 1e-3) / 1e9, 2) if stage_ms > 0 else None,
                frac=round(alg_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if stage_ms > 0 else None)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process is only a parent.  It has made no GPU call
    (no torch.cuda, no libhybkf) and starts N fresh rank processes -- `python -m torch.distributed.run`, one per GPU, rendezvous on
    127.0.0.1 -- waits for them and exits with their status.  It never exec()s."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def collective_selftest(args):
    """No GPU needed: the ranks form a gloo group and run SlabExchange -- the very sequence of collectives SlabPipeline issues per
    frame (MIN all-reduce of the 64-bit crossing words, normals by the vertex's owner, integer SUM all-reduce, unpack) -- on CPU tensors, with plain-torch
    restatements of the device launches (tests/slab_cpu_ops.py), and check the merged maps against the first-crossing rule.
    Used by tests/test_slab_distributed_cpu.py to cover the launcher path end to end at world_size 2."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import slab_cpu_ops as ops
    from hybkinectfu_amd.pipeline import SlabExchange
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    rows, cols = 48, 64
    model, step = {}, {}
    ex = SlabExchange(rows, cols, torch.device("cpu"), dist, normals=lambda ta, cand: step["normals"](ta, cand),
                      unpack=lambda ta, cand: model.update(zip("vn", ops.unpack(ta, cand))))
    ok, overlapped = True, 0
    for frame in range(3):
        ta, normals, want_ta, want_cand, want_v, want_n = ops.synthetic_crossings(rows, cols, rank, world, seed=100 + frame)
        step["normals"] = normals
        ex.ta.copy_(ta)

        def overlap():
            nonlocal overlapped
            overlapped += 1
        ex.merge(overlap)
        ok = ok and torch.equal(model["v"].view(torch.int32), want_v.view(torch.int32)) and torch.equal(model["n"].view(torch.int32), want_n.view(torch.int32))
    flag = torch.tensor([1 if ok and overlapped == 3 else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps(dict(selftest="slab-collectives", n_gpus=args.gpus, world_size=dist.get_world_size(), backend="gloo",
                              frames=3, ok=bool(flag.item()))))
    dist.destroy_process_group()
    return 0 if flag.item() else 1


def pcie_inclusive(wl, frames_mm, n_frames=100, warmup=10, repeats=3):
    """Frames/s when every frame starts in HOST memory: SingleGpuPipeline.process_frame_host -- pinned staging, DMA on a copy stream, frame k + 2
    crossing PCIe (kf_upload_depth_mm_next, a free upload slot) while frame k is processed and frame k + 1's front end rides in frame k's launches.
    Every frame is uploaded inside the timed region.  Reported beside `value`, never as `value`.  The rate depends on where the runtime places the
    pinned staging buffers (it differs from context to context on the same box), so `repeats` fresh contexts are timed: value = their median."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    cam = wl["cam"]
    host = [np.ascontiguousarray(f, np.uint16) for f in frames_mm]

    def frame_of(k):
        return host[k % len(host)]
    rates = []
    for rep in range(repeats):
        pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=torch.cuda.current_device())
        # the first host-fed context of a PROCESS pays ~25 ms of one-time set-up in the upload path (pinned staging + the copy stream's first DMAs: its first ~35
        # frames run at half rate, profiles/r05_pcie_first_run.txt): it warms up for 60 frames, the later ones for `warmup`
        warm = max(warmup, 60) if rep == 0 else warmup
        for k in range(warm):
            pipe.process_frame_host(frame_of, k)
        pipe.sync()
        t0 = time.perf_counter()
        for k in range(warm, warm + n_frames):
            pipe.process_frame_host(frame_of, k)
        pipe.sync()
        rates.append(round(n_frames / (time.perf_counter() - t0), 2))
        pipe.close()
    return dict(value=sorted(rates)[len(rates) // 2], unit="frames/s", steps=n_frames, runs=rates, bytes_per_frame=int(cam[0] * cam[1] * 2),
                note="every frame uploaded from host memory over PCIe inside the timed region, two frames ahead of its use (kf_upload_depth_mm_next + "
                     "kf_prefetch_frame); median of %d fresh contexts (the first one of the process warms up for 60 frames: a one-time ~25 ms of set-up in the upload path, "
                     "profiles/r05_pcie_first_run.txt)" % repeats)


def stats_every_frame_block(wl, n_frames=150, warmup=10):
    """A host that asks for the observed-voxel count after EVERY frame, as the reference prints it (src/cuda/integrateVolume.cu:91-94): kf_get_volume_stats is a
    blocking read-back per frame; from the second question on the fusion launches keep the count themselves (their COUNT instantiations), so no call sweeps the
    volume (round 4: a 0.25 ms sweep per call at 512^3, 1.5 ms at 1024^3).  The last answer is cross-checked against a sweep (kf_count_observed_voxels)."""
    import torch
    from hybkinectfu_amd.pipeline import SingleGpuPipeline
    cam = wl["cam"]
    frames, _ = S.make_stream(100, cam, wl["size"])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SingleGpuPipeline(K.camera(*cam), wl["res"], wl["size"], wl, device=torch.cuda.current_device())
    count = 0
    for k in range(warmup):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
        count = pipe.stats(observed=True)["weight_gt0"]
    pipe.sync()
    t0 = time.perf_counter()
    for k in range(warmup, warmup + n_frames):
        pipe.process_frame_device(dev.data_ptr() + (k % 100) * fb, k, dev.data_ptr() + ((k + 1) % 100) * fb)
        count = pipe.stats(observed=True)["weight_gt0"]
    pipe.sync()
    dt = time.perf_counter() - t0
    swept = pipe.ctx.count_observed_voxels()
    pipe.close()
    return dict(workload=wl["desc"], value=round(n_frames / dt, 2), unit="frames/s", steps=n_frames, observed_voxels=int(count), swept=int(swept), equal=bool(count == swept),
                note="kf_get_volume_stats (blocking read-back) after every frame; the count is kept by the fusion launches, not swept (profiles/r05_observed_count.txt: "
                     "2.1 k frames/s with a sweep per call)")


def profiled_traffic(key):
    """HBM-side bytes per launch from the builder's rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, profiles/integrate_traffic.json), or None"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "integrate_traffic.json")))
        src = tj.get("source") or ""
        cited = src.split(":", 1)[0].strip()
        if cited.startswith("profiles/") and not os.path.exists(os.path.join(ROOT, cited)):
            src = "(cited file %s is missing from the tree) %s" % (cited, src)       # a provenance pointer must resolve
        return tj.get(key), src
    except Exception:
        return None, None


def roofline_extra(pipe, run, first_frame, res, size, n_frames=20, wl_name="C2"):
    """Raycast and marching cubes against the HBM roofline, as SURVEY.md section 8d defines their bytes.
    raycast: what the REFERENCE's march would read -- 8 B per sample from t_min to the first crossing (or t_max) plus 64 voxels
    x 8 B per evaluated hit -- counted on the device (kf_read_work_counters) over n_frames extra frames, / the kernel's own time.
    The HIP kernel skips empty space through its bit tables, so its real traffic is far below this figure: the quotient says
    how fast the reference's work is DONE, not how busy HBM is (the march is latency-bound).
    marching cubes: TOUCHED bytes -- bricks whose 3x3x3 brick neighbourhood holds a negative voxel x 4 KiB (what the extraction
    reads, once) + 72 B per triangle -- / the extraction's time (all its kernels: dilate ... emit)."""
    c = pipe.ctx
    c.stage_timers((1 << 7) | (1 << 6) | (1 << 16))
    run(first_frame, n_frames)
    pipe.sync()
    ms, cnt = c.read_stage_ms()
    steps, hits, _, _ = c.work_counters()
    rc_ms = float(ms[7]) / max(int(cnt[7]), 1)
    rc_bytes = (steps * 8.0 + hits * 64 * 8.0) / n_frames
    rc_traffic, tsrc = profiled_traffic(wl_name + "_raycast")
    out = dict(raycast=dict(kernel="k_raycast", bound="latency (L2 gathers)", ms=round(rc_ms, 5), reference_samples_per_frame=int(steps / n_frames),
                            hits_per_frame=int(hits / n_frames), reference_bytes_per_launch=int(rc_bytes),
                            reference_bytes_rate=round(rc_bytes / (rc_ms * 1e-3) / 1e9, 2) if rc_ms > 0 else None, unit="GB/s",
                            reference_bytes_rate_over_peak=round(rc_bytes / (rc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if rc_ms > 0 else None,
                            kernel_traffic_bytes_per_launch=rc_traffic,
                            kernel_traffic_rate=round(rc_traffic / (rc_ms * 1e-3) / 1e9, 2) if (rc_traffic and rc_ms > 0) else None,
                            kernel_traffic_frac_of_hbm_peak=round(rc_traffic / (rc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (rc_traffic and rc_ms > 0) else None,
                            kernel_traffic_source=tsrc if rc_traffic else None,
                            note="reference_bytes = what the REFERENCE's march reads (8 B per sample + 64 voxels per hit); the kernel skips empty space through its "
                                 "bit tables and moves far less: reference_bytes_rate is not an HBM rate"))
    c.marching_cubes(300.0 * size / res)             # untimed first call: allocates the extraction's scratch buffers
    c.clear_triangles()
    c.stage_timers((1 << 6) | (1 << 16))
    c.marching_cubes(300.0 * size / res)
    pipe.sync()
    ms, cnt = c.read_stage_ms()
    _, _, bricks, tris = c.work_counters()
    mc_ms = float(ms[6]) / max(int(cnt[6]), 1)
    mc_bytes = bricks * 4096.0 + tris * 72.0       # bricks whose 3x3x3 brick neighbourhood holds a negative voxel x 4 KiB + the triangles
    out["marching_cubes"] = dict(kernels="k_mc_dilate + k_mc_codes + k_mc_sift + k_mc_list + k_mc_count + k_mc_scan_* + k_mc_emit_recs", bound="latency / launch (11 short kernels)",
                                 ms=round(mc_ms, 5), bricks_read=int(bricks), bricks_total=int((res // 8) ** 3), triangles=int(tris), touched_bytes=int(mc_bytes),
                                 dense_bytes=int(res ** 3 * 8 + tris * 72),
                                 touched_bytes_rate=round(mc_bytes / (mc_ms * 1e-3) / 1e9, 2) if mc_ms > 0 else None, unit="GB/s",
                                 touched_bytes_rate_over_peak=round(mc_bytes / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mc_ms > 0 else None,
                                 dense_reference_bytes_rate_over_peak=round((res ** 3 * 8 + tris * 72) / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mc_ms > 0 else None,
                                 note="touched_bytes = bricks read x 4 KiB + 72 B per triangle (counted on the device); dense_reference_bytes = R^3 x 8 B + triangles, what the "
                                      "reference's dense sweep reads -- neither is a measured HBM rate")
    mc_traffic, tsrc = profiled_traffic(wl_name + "_marching_cubes")
    out["marching_cubes"].update(kernel_traffic_bytes_per_extraction=mc_traffic,
                                 kernel_traffic_rate=round(mc_traffic / (mc_ms * 1e-3) / 1e9, 2) if (mc_traffic and mc_ms > 0) else None,
                                 kernel_traffic_source=tsrc if mc_traffic else None)
    c.clear_triangles()
    c.stage_timers(0)
    return out


def per_rank_leg(pipe, run, barrier, dist, world, rank, first, n_frames):
    """z-slab runs: what every rank did over n_frames extra frames (outside the timed region) -- voxels fused, bricks queued, and the
    device time of its stages (HIP events) plus the merge (torch events around both all-reduces, mask, unpack; the next frame's
    preprocess is enqueued inside it while the first all-reduce is in flight).  Slabs are not equally busy: this makes it visible."""
    import torch
    pipe.stage_timers((1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5))
    pipe.time_merge(True)
    s0 = pipe.stats()
    run(first, n_frames)
    barrier()
    ms, cnt = pipe.read_stage_ms()
    merge_ms, merge_n = pipe.read_merge_ms()
    pipe.time_merge(False)
    pipe.stage_timers(0)
    s1 = pipe.stats()
    us = lambda i: 1000.0 * float(ms[i]) / max(int(cnt[i]), 1)
    mine = torch.tensor([float(rank), (s1["updated_total"] - s0["updated_total"]) / float(n_frames), float(s1["bricks_active"]), us(1), us(2), us(3), us(5), us(4),
                         1000.0 * merge_ms / max(merge_n, 1), float(s1["frames_lost"] - s0["frames_lost"]), float(pipe.slab[0]), float(pipe.slab[1])],
                        device="cuda", dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    keys = ["rank", "n_upd_per_frame", "bricks_queued_last_frame", "preprocess_us", "track_us", "integrate_us", "integrate_kernel_us", "raycast_us", "merge_us",
            "frames_lost", "z_begin", "z_end"]
    rows = [{k: (int(v) if k in ("rank", "n_upd_per_frame", "bricks_queued_last_frame", "frames_lost", "z_begin", "z_end") else round(float(v), 2))
             for k, v in zip(keys, e.tolist())} for e in every]
    return dict(frames=n_frames, ranks=rows,
                note="device time per frame and rank (HIP events; raycast: the marching launch incl. the speculative normals of the rank's own crossings; "
                     "merge: torch events around MIN all-reduce + normals (a copy where the rank's crossing won) + integer SUM all-reduce + unpack, "
                     "all enqueued on the pipeline's own stream)")


def balanced_ranges(args, kcam, res, size, wl, world, device, first_frame_ptr):
    """z-slab boundaries of a run: equal thickness, or (default for N > 1) chosen so that the busiest rank fuses as little as possible, from a
    one-frame probe of the work per z-layer in a 256^3 volume of the same extent (pipeline.probe_layer_work; every rank computes the same)."""
    from hybkinectfu_amd import pipeline as PL
    if world == 1 or args.slab_balance == "equal":
        return None
    inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
    work = PL.probe_layer_work(kcam, res, size, wl, first_frame_ptr, device=device)
    return PL.slab_ranges(res, world, work, halo=PL.slab_halo_layers(res, size, inc))


def slab_block(name, args, world, rank, device, dist, n_frames, warmup, n_unique):
    """A short run of workload `name` on the z-slab pipeline of THIS process group (the N > 1 line's C5 block): frames/s over
    n_frames - warmup frames, per-rank statistics, lock-step check, and the mesh extraction of every rank's slab."""
    import ctypes as C
    import torch
    from hybkinectfu_amd.pipeline import SlabPipeline
    wl = workload(world, name)
    cam, res, size = wl["cam"], wl["res"], wl["size"]
    frames, _ = S.make_stream(n_unique, cam, size)
    dev_frames = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    pipe = SlabPipeline(K.camera(*cam), res, size, wl, rank=rank, world=world, device=device, icp_mode=args.icp_mode, tracker=args.tracker,
                        max_triangles=(16_000_000 // world + 1_000_000 if wl.get("extract_mesh") else 0),
                        ranges=balanced_ranges(args, K.camera(*cam), res, size, wl, world, device, dev_frames.data_ptr()))

    def run(first, count):
        for k in range(first, first + count):
            pipe.process_frame_device(dev_frames.data_ptr() + ping_pong(k, n_unique) * fb, k, dev_frames.data_ptr() + ping_pong(k + 1, n_unique) * fb)

    def barrier():
        pipe.sync(); torch.cuda.synchronize(); dist.barrier()

    run(0, warmup)
    barrier()
    s0 = pipe.stats()
    t0 = time.perf_counter()
    run(warmup, n_frames - warmup)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    lost = pipe.stats()["frames_lost"] - s0["frames_lost"]
    out = dict(workload=wl["desc"], value=round((n_frames - warmup) / dt, 2), unit="frames/s", steps=n_frames - warmup, warmup=warmup,
               ms_per_step=round(1000.0 * dt / (n_frames - warmup), 4), frames_lost=int(lost), halo_layers=int(pipe.halo),
               slab_ranges=[list(r) for r in pipe.ranges], slab_balance=args.slab_balance,
               unique_frames=n_unique, per_rank=per_rank_leg(pipe, run, barrier, dist, world, rank, n_frames, 10))
    out["lockstep"] = pipe.verify_lockstep()
    if wl.get("extract_mesh"):
        barrier()
        t0m = time.perf_counter()
        pipe.ctx.marching_cubes(300.0 * size / res)
        n_tri = C.c_uint32()
        K._chk(pipe.ctx.lib.kf_triangle_count(pipe.ctx.h, C.byref(n_tri)), "kf_triangle_count")
        barrier()
        tt = torch.tensor([float(n_tri.value)], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        out["mesh_extraction"] = dict(triangles=int(tt.item()), ms=round(1000.0 * (time.perf_counter() - t0m), 3),
                                      note="kf_marching_cubes on every rank's slab (first call: includes the scratch allocation), wall time incl. the count read-back")
    pipe.close()
    return out


STAGE_NAMES = ["upload", "preprocess", "track", "integrate", "raycast", "integrate_kernel", "mcubes", "raycast_kernel"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # warmup + steps stay below max_weight = 128 fused frames: one regime
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs of the N=1 line (per-stage pass, raycast / marching-cubes rooflines, PCIe-inclusive rate)")
    ap.add_argument("--pre-frames", type=int, default=0,
                    help="frames fused before the warm-up (e.g. 160 = past max_weight: the timed frames then run the saturation-aware kernels; "
                         "used to profile that regime -- the default 0 keeps warm-up + steps inside the early regime)")
    ap.add_argument("--tracker", default="icp", choices=["icp", "sdf"], help="z-slab pipeline only: CameraPoseFinderICP (default) or CameraPoseFinderSDF on slabs")
    ap.add_argument("--slab-balance", default="probe", choices=["probe", "equal"],
                    help="N > 1: slab boundaries from a one-frame low-resolution probe of the work per z-layer (default: the busiest rank sets the frame time, "
                         "and equal z-slabs are unequally busy) or equal thickness")
    ap.add_argument("--rebalance-every", type=int, default=0,
                    help="N > 1: every K frames the ranks pool the work per brick layer counted by the fusion pass, re-run the boundary optimisation and move the layers "
                         "that change owner rank to rank (SlabPipeline.rebalance); 0 (default): the boundaries of the start-up probe stay -- the benchmark trajectory is a "
                         "2 cm circle, its work does not move along z")
    ap.add_argument("--no-c5", action="store_true", help="N > 1: skip the short C5 block (2048^3 @ 8 m, 1280x960) that follows the C4 measurement")
    ap.add_argument("--config", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5"],
                    help="auto: C2 (512^3 @ 4 m) on 1 GPU, C4 (1024^3 @ 6 m, z-slabs) on N > 1, as BASELINE.json's metric states; "
                         "c5: 2048^3 @ 8 m, 1280x960 depth, mesh extraction after the timed frames (BASELINE.json configs[4])")
    ap.add_argument("--icp-mode", default="replicated", choices=["replicated", "allreduce"],
                    help="multi-GPU tracking: every rank runs the whole ICP (default) or pixels are split and the 27-float system all-reduced")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="skip the extra 50-frame run of the multi-GPU workload (C4) on this one GPU that the N=1 line reports beside the headline")
    ap.add_argument("--prefetch", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not hand the pipeline the next frame's address: by default frame k+1's depth conversion + gate + bilateral filter ride in frame k's "
                         "raycast launch (kf_prefetch_frame, fused form) -- the frames are a resident stream, the work per frame is the same, one frame earlier")
    ap.add_argument("--force-slab", action="store_true", help="run the z-slab pipeline (and its collectives) even with one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL, one GPU per rank (the measured configuration).  gloo: rehearsal only -- ranks may SHARE a GPU "
                         "(device = local_rank %% device_count), collectives staged through host memory")
    ap.add_argument("--collective-selftest", action="store_true",
                    help="CPU only: run the z-slab pipeline's collective sequence over gloo on synthetic candidates and print one JSON line")
    args = ap.parse_args()

    # ---- launcher: a parent that has touched neither the GPU nor the library starts the N ranks ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.collective_selftest:
        sys.exit(collective_selftest(args))

    global K
    import torch
    from hybkinectfu_amd import lib as K
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit("--gpus %d needs %d GPUs, %d visible (RCCL wants one GPU per rank; --backend gloo rehearses ranks sharing a GPU)" % (world, world, n_dev))
    device = local_rank % n_dev
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or args.force_slab:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    wl = workload(world, args.config)
    if args.tracker == "sdf":
        wl = dict(wl, tracker="sdf")
    cam, res, size = wl["cam"], wl["res"], wl["size"]
    kcam = K.camera(*cam)
    period = 100
    n_unique = min(period, args.pre_frames + args.warmup + args.steps)
    frames, _ = S.make_stream(n_unique, cam, size)
    dev_frames = torch.from_numpy(frames.astype(np.int16)).cuda()        # u16 bits, resident in HBM
    frame_bytes = cam[0] * cam[1] * 2

    slab = not (world == 1 and not args.force_slab)
    if not slab:
        from hybkinectfu_amd.pipeline import SingleGpuPipeline as Pipe
        pipe = Pipe(kcam, res, size, wl, device=device, max_triangles=(16_000_000 if wl.get("extract_mesh") else 4_000_000 if not args.no_extras else 0))
    else:
        from hybkinectfu_amd.pipeline import SlabPipeline as Pipe
        pipe = Pipe(kcam, res, size, wl, rank=rank, world=world, device=device, icp_mode=args.icp_mode, tracker=args.tracker,
                    max_triangles=(16_000_000 // world + 1_000_000 if wl.get("extract_mesh") else 0),
                    ranges=balanced_ranges(args, kcam, res, size, wl, world, device, dev_frames.data_ptr()), rebalance_every=args.rebalance_every)

    def run(first, count):
        # --prefetch: frame k+1 is preprocessed on the context's side stream while frame k is tracked (kf_prefetch_frame)
        for k in range(first, first + count):
            # (the z-slab pipeline always gets the hint: it fills the wait for its first all-reduce with the next frame's preprocess)
            nxt = dev_frames.data_ptr() + ((k + 1) % n_unique) * frame_bytes if (slab or not args.no_prefetch) else None
            pipe.process_frame_device(dev_frames.data_ptr() + (k % n_unique) * frame_bytes, k, nxt)

    def barrier():
        pipe.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    base = args.pre_frames
    if base:
        run(0, base)
    run(base, args.warmup)
    barrier()
    s0 = pipe.stats()
    # Time the fusion kernel (bit 5) and the whole integrate stage (bit 3) inside the timed region with HIP events on the context's
    # own stream.  Every `period`-th frame is sampled so that a short run still yields >= 5 timed launches (a timed launch costs
    # ~6 us of stream time -- its dispatch carries two event packets: every 2nd frame timed takes 1.4 % off `value` at C2, every 4th
    # 0.7 %, every 8th nothing measurable; the kernel's duration varies by < 2 % from launch to launch).
    timer_period = max(1, min(8, args.steps // 5))
    # (every event pair costs ~3 us of stream time: at --steps 20 the kernel's pair alone is 0.9 % of the timed region, the stage's pair
    # another 1.4 % -- so when the per-stage times are measured anyway on the 50 frames AFTER the timed region (N=1 extras), the
    # integrate STAGE is taken from there and only the roofline kernel is timed inside the region)
    extras_planned = world == 1 and not wl.get("extract_mesh") and not args.force_slab and not args.no_extras
    pipe.stage_timers((timer_period << 8) | (1 << 5) | (0 if extras_planned else (1 << 3)))
    barrier()
    t0 = time.perf_counter()
    run(base + args.warmup, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms, cnt = pipe.read_stage_ms()
    s1 = pipe.stats()
    lost = s1["frames_lost"] - s0["frames_lost"]
    n_upd = (s1["updated_total"] - s0["updated_total"])
    if dist is not None:
        t = torch.tensor([float(n_upd), float(lost)], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_upd_all, lost = int(t[0].item()), int(t[1].item()) // world
    else:
        n_upd_all = n_upd
    fps = args.steps / dt

    # ---- roofline of the dominant HBM kernel: the fusion pass k_integrate_pairs ----
    # The timed region runs the kernel with deferred free-space weights (DEFER): it provably does not move the reference's bytes, so the
    # reference's bytes over ITS time is not an HBM rate.  The roofline fraction is therefore measured on the PLAIN read-modify-write form of
    # the same kernel (kf_set_defer(0)), live, on the frames that follow the timed region in the same pipeline: algorithmic bytes (N_upd x 16 B
    # + the depth image) / the kernel's own dispatch time.  The DEFER kernel of the timed region is reported beside it
    # with what it really moves (PMC traffic from the builder's rocprofv3 passes) over its live time.
    defer_launches = int(cnt[5])
    defer_ms = float(ms[5]) / max(defer_launches, 1)
    stage_ms = float(ms[3]) / max(int(cnt[3]), 1)
    n_upd_region = n_upd
    sat_env = os.environ.get("KF_INTEGRATE_SAT", "1")      # (kf_defer_enabled, integrate.hip: by default volumes of 768^3 and finer defer)
    deferral_on = (sat_env == "2" or (sat_env == "1" and res >= 768)) and os.environ.get("KF_INTEGRATE_PAIRS", "1") != "0" and not wl.get("color")
    plain_frames = 24 if res <= 1024 else 12
    pipe.ctx.set_defer(0)
    run(base + args.warmup + args.steps, 2)                 # the first plain launch is preceded by the flush of every pending count
    barrier()
    sp0 = pipe.stats()
    pipe.stage_timers((2 << 8) | (1 << 5))
    run(base + args.warmup + args.steps + 2, plain_frames)
    barrier()
    pms, pcnt = pipe.read_stage_ms()
    sp1 = pipe.stats()
    pipe.stage_timers(0)
    pipe.ctx.set_defer(-1)
    run(base + args.warmup + args.steps + 2 + plain_frames, 3)     # the deferred states are re-established before any other leg
    barrier()
    after_roofline = base + args.warmup + args.steps + 5 + plain_frames
    launches = int(pcnt[5])
    kern_ms = float(pms[5]) / max(launches, 1)
    n_upd = (sp1["updated_total"] - sp0["updated_total"]) / float(plain_frames)
    roof_rank = 0
    if dist is not None and world > 1:
        # z-slabs are not equally busy (the camera's near slabs see a narrow frustum): quote the rank that fuses the most voxels
        mine = torch.tensor([float(n_upd), kern_ms, float(launches), stage_ms, defer_ms, float(defer_launches), float(n_upd_region)], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        roof_rank = int(max(range(world), key=lambda r: float(every[r][0])))
        e = every[roof_rank]
        n_upd, kern_ms, launches, stage_ms, defer_ms, defer_launches, n_upd_region = (float(e[0].item()), float(e[1].item()), int(e[2].item()), float(e[3].item()),
                                                                                       float(e[4].item()), int(e[5].item()), float(e[6].item()))
    alg_bytes = n_upd * 16.0 + cam[0] * cam[1] * 4.0       # N_upd x 2 x 8 B + depth map
    alg_region = (n_upd_region / max(args.steps, 1)) * 16.0 + cam[0] * cam[1] * 4.0
    if launches >= 5 and kern_ms > 0:
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = profiled_traffic(wl["name"]) if world == 1 else (None, None)
        roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=traffic, traffic_source=traffic_source, kernel="k_integrate_pairs (plain read-modify-write form: deferral off)", kernel_ms=round(kern_ms, 5),
                        launches_timed=int(launches), rank=roof_rank, algorithmic_bytes_per_launch=int(alg_bytes), n_upd_per_frame=int(n_upd),
                        measured="live in this run, HIP events stamped by the kernel's own dispatch, every 2nd of the %d frames that follow the timed region in the same "
                                 "pipeline with kf_set_defer(0) (the timed region itself runs the DEFER form: in_timed_region)" % plain_frames)
        dtraffic, dsrc = profiled_traffic(wl["name"] + "_deferred") if world == 1 else (None, None)
        roofline["in_timed_region"] = dict(
            kernel="k_integrate_pairs<.., DEFER>" if deferral_on else "k_integrate_pairs (plain)", kernel_ms=round(defer_ms, 5), launches_timed=int(defer_launches),
            reference_bytes_per_launch=int(alg_region),
            reference_bytes_rate=round(alg_region / (defer_ms * 1e-3) / 1e9, 2) if defer_ms > 0 else None,
            kernel_traffic_bytes_per_launch=dtraffic,
            kernel_traffic_rate=round(dtraffic / (defer_ms * 1e-3) / 1e9, 2) if (dtraffic and defer_ms > 0) else None,
            kernel_traffic_frac_of_hbm_peak=round(dtraffic / (defer_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (dtraffic and defer_ms > 0) else None,
            kernel_traffic_source=dsrc if dtraffic else None, unit="GB/s",
            stage=dict(kernels="whole integrate stage: k_integrate_pairs, plus k_integrate_cull where it is a launch of its own (at C2 the cull runs as the tail of the tracking launch)", ms=round(stage_ms, 5)) if stage_ms > 0 else None,
            note="whole free-space quarter bricks are counted, not read or written (deferred weights): reference_bytes_rate is the reference's bytes over this "
                 "kernel's time, NOT an HBM rate; kernel_traffic_* is what the kernel moves (PMC, 2 x FETCH_SIZE + WRITE_SIZE)")
    else:
        roofline = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None, kernel="k_integrate_pairs",
                        launches_timed=int(launches), refused="fewer than 5 timed launches of the kernel")

    mesh = None
    if wl.get("extract_mesh"):
        # C5: every rank extracts its own slab (slab-major concatenation = the single-GPU (z, y, x, k) order); outside the timed frames
        import ctypes as C
        barrier()
        t0m = time.perf_counter()
        pipe.ctx.marching_cubes(300.0 * size / res)
        n_tri = C.c_uint32()
        K._chk(pipe.ctx.lib.kf_triangle_count(pipe.ctx.h, C.byref(n_tri)), "kf_triangle_count")
        barrier()
        dtm = time.perf_counter() - t0m
        total = n_tri.value
        if dist is not None:
            tt = torch.tensor([float(total)], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            total = int(tt.item())
        mesh = dict(triangles=total, ms=round(1000.0 * dtm, 3), note="kf_marching_cubes on every rank's slab after the timed frames (wall time incl. the count read-back)")

    n_before = base + args.warmup
    maxw = int(P["volume_max_weight"])
    out = dict(metric="depth frames/sec into TSDF (integrate+ICP+raycast)", value=round(fps, 2), unit="frames/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(1000.0 * dt / args.steps, 4), higher_is_better=True,
               scaling="weak" if world == 1 else "strong", vs_baseline=None, dtype="f32", data="synthetic",
               value_resident=round(fps, 2),
               deferral=dict(enabled=deferral_on, frames_fused_before_timed=n_before, timed_frames=args.steps, max_weight=maxw,
                             note="deferred free-space weights (integrate.hip, DEFER): from the second fused frame on, a quarter brick whose 128 voxels hold tsdf 1 and are "
                                  "all observed as free space again is counted, not read or written; bit-identical results (tests/test_gpu_saturation.py, "
                                  "tools/sat_equivalence.py).  Default: on for volumes of 768^3 and finer (1024^3 +46 %, 2048^3 +86 % frames/s), off below (512^3 with "
                                  "the stock 2 m gate: -1.5 %, the fusion pass is not memory-bound there); kf_set_defer / KF_INTEGRATE_SAT=0|2 override"),
               config=dict(workload=wl["desc"], volume="%d^3 @ %g m" % (res, size), image="%dx%d" % (cam[0], cam[1]),
                           tracker=("CameraPoseFinderSDF (max %d iterations, device-resident)" % int(P["sdf_max_iter_nums"])
                                    if (wl.get("tracker") == "sdf" or (slab and args.tracker == "sdf")) else "ICP 10/5/4 (device-resident Gauss-Newton)"),
                           frames_lost=int(lost), env=env_knobs(),
                           world_size=(dist.get_world_size() if dist is not None else 1),
                           backend=("none" if dist is None else ("rccl" if args.backend == "nccl" else "gloo (rehearsal: host-staged collectives, ranks may share a GPU)")),
                           partition="none" if not slab else
                           "z-slab x%d (boundaries: %s); %d halo layers per side RE-INTEGRATED by both neighbours (recomputed, not exchanged per frame; when the boundaries move -- "
                           "--rebalance-every -- whole brick layers travel rank to rank: SlabMigrator); "
                           "raycast merge = MIN all-reduce (crossing parameter + vertex parameter as one 64-bit word, 2.5 MB at VGA) + integer SUM all-reduce (normal from the vertex's owner as three words, 3.7 MB); ICP %s"
                           % (world, "balanced from a one-frame 256^3 probe of the work per z-layer" if (world > 1 and args.slab_balance == "probe") else "equal thickness",
                              pipe.halo, args.icp_mode),
                           slab_ranges=([list(r) for r in pipe.ranges] if slab else None)),
               roofline=roofline)
    if mesh is not None:
        out["mesh_extraction"] = mesh
    if slab and dist is not None:
        # z-slab runs: per-rank statistics over 20 extra frames, then the lock-step check (all ranks fused / lost the same frames, same pose bits)
        out["per_rank"] = per_rank_leg(pipe, run, barrier, dist, world, rank, after_roofline, 20)
        out["lockstep"] = pipe.verify_lockstep()
        out["slab_migrations"] = [dict(frame=f, old=[list(r) for r in o], new=[list(r) for r in n], voxel_layers_moved_by_rank0=m) for f, o, n, m in pipe.migrations]
        # plans that promised the gain threshold but would not have paid for their wire time before the next decision (pipeline.migration_pays)
        out["slab_migrations_declined"] = [dict(frame=f, gain_ms=round(g * 1e3, 3), cost_ms=round(c * 1e3, 3)) for f, g, c in pipe.declined]
        if world > 1 and args.config == "auto" and not args.no_c5:
            # the north star quotes 1024^3 AND 2048^3 for 1/2/4/8 GPUs: a short C5 block (2048^3 @ 8 m, 1280x960 depth, mesh extraction) on the same ranks
            pipe.close()
            out["c5"] = slab_block("c5", args, world, rank, device, dist, n_frames=30, warmup=6, n_unique=12)
    extras = world == 1 and not wl.get("extract_mesh") and not args.force_slab and not args.no_extras
    if extras:
        # per-stage device time (HIP events around every stage, 50 extra frames outside the timed region)
        pipe.stage_timers(0x1F | (1 << 5))
        run(after_roofline, 50)
        pipe.sync()
        sm, sc = pipe.read_stage_ms()
        out["stage_us"] = {STAGE_NAMES[i]: round(1000.0 * float(sm[i]) / max(int(sc[i]), 1), 2) for i in (1, 2, 3, 4, 5)}
        out["stage_us"]["note"] = "mean device time per frame, HIP events around each stage, 50 frames after the timed region"
        if out["roofline"].get("frac") is not None:
            out["roofline"]["stage"] = stage_block(out["roofline"]["algorithmic_bytes_per_launch"], float(sm[3]) / max(int(sc[3]), 1),
                                                   "HIP events on the 50 frames after the timed region (inside it only the kernel is timed)")
        pipe.stage_timers(0)
        out["roofline_extra"] = roofline_extra(pipe, run, after_roofline + 50, res, size, wl_name=wl["name"])
    if world == 1 and args.config == "auto" and not args.force_slab and not args.no_scaling_reference:
        pipe.close()
        out["multi_gpu_workload_on_1_gpu"] = single_gpu_reference("c4")      # what --gpus 2/4/8 should be compared with
    if extras and args.config == "auto":
        # the other BASELINE.json configurations next to the headline (short runs, each on a fresh context), and C2 with the reference's
        # stock colour switches (use_color=1, color_angle_weight=1: src/config.ini:2,7)
        pipe.close()
        out["other_configs"] = {
            "C1": single_gpu_reference("c1", n_frames=70),
            "C3": single_gpu_reference("c3", n_frames=70),
            "C5_on_1_gpu": single_gpu_reference("c5", n_frames=30, warmup=6, n_unique=12),
            "C2_color": single_gpu_reference("c2", n_frames=110, overrides=dict(color=True, desc="C2 with use_color=1, color_angle_weight=1 (the reference's stock switches): "
                                                                                 "colour planes fused and raycast beside tsdf / weight; one synthetic colour image reused")),
        }
    if extras:
        pipe.close()
        out["pcie_inclusive"] = pcie_inclusive(wl, frames)
        # the metric's own wording includes the upload ("upload + preprocess + track + integrate + raycast"): both rates side by side
        out["value_pcie_inclusive"] = out["pcie_inclusive"]["value"]
        out["value_note"] = ("value = value_resident: frames already in HBM when the timed region starts (the bench contract); value_pcie_inclusive: every frame "
                             "crosses PCIe inside the timed region (kf_upload_depth_mm / kf_upload_depth_mm_next): SURVEY.md section 8d's metric includes the upload, so the "
                             "metric's LITERAL value is value_pcie_inclusive")
        out["steady_state"] = {"C2": steady_state("c2"), "C4": steady_state("c4")}
        if args.config == "auto":
            out["scene_noise"] = scene_noise_block(wl)
            out["stats_every_frame"] = stats_every_frame_block(wl)
    if world == 1 and not args.no_cpu_baseline and not wl.get("extract_mesh"):      # (the oracle's 2048^3 volume would need 103 GB of host memory)
        pipe.close()
        n_sample = 150 if wl["res"] <= 512 else 40                                    # 10-20 s of CPU work either way
        out["cpu_baseline"], out["parity_witness"] = cpu_baseline(wl, frames, n_sample=n_sample, witness=gpu_witness_run(wl, frames, n_sample))
        # top level, so that a failing witness cannot hide in a nested block (BENCH_r04 said `false` and nobody noticed)
        out["parity_witness_ok"] = bool(out["parity_witness"]["pose_within_tolerance"] and out["parity_witness"]["n_upd_equal"])
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
