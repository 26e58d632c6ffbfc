#!/usr/bin/env python3
"""Micro-benchmark of the ICP tracker alone on the C2 stream: fuse 3 frames, then track frame 3 repeatedly from the same pose."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
P = S.STOCK
cam, size, res = S.vga_camera(), 4.0, 512
ctx = K.Context(K.camera(*cam), res, size, P["volume_max_weight"], levels=3)
ctx.set_pose(S.pose0(size))
for k in range(4):
    ctx.upload_depth_mm(S.render_depth_mm(S.trajectory_pose(k, size), cam, size))
    ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
    if k < 3:
        ctx.integrate(None, P["integrate_sdf_trunc"], P["integrate_depth_trunc"])
        ctx.raycast(None, 0.035, P["depth_trunc_min"], P["depth_trunc_max"])
ok, pose, status, iters = ctx.track_result()
print("tracked", ok, "status", status, "iters", iters)
ctx.stage_timers(1 << 2)
for _ in range(50):
    ctx.icp_track(4, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
ms, cnt = ctx.read_stage_ms()
print("track stage: %.4f ms per frame (pyramids + begin + loop)" % (ms[2] / cnt[2]))
if os.environ.get("KF_ICP_EXP") == "7":
    r = ctx.read_solver_params()
    names = ["prefetch", "fold", "solve", "pixels", "wavesum+store", "barrier"]
    print("WG0 lane0 shader ticks per frame: " + ", ".join("%s=%.0f" % (n, r[20 + i]) for i, n in enumerate(names)), "(100 MHz ticks? see s_memtime)")
if os.environ.get("KF_ICP_EXP") == "8":
    # per-workgroup wall-clock stamps (s_memrealtime, 10 ns ticks): [step][wg] -> (published its partial, finished folding everyone's)
    import ctypes as C
    buf = np.zeros(19 * 1024, np.uint64)
    ctx.lib.kf_exp_read_icp_slots(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(24 * 1024 * 32), C.c_size_t(buf.size))
    b = buf.reshape(19, 512, 2).astype(np.int64)
    grids = [13] * 4 + [50] * 5 + [200] * 10
    for s_, g in enumerate(grids):
        pub, fold = b[s_, :g, 0], b[s_, :g, 1]
        t0 = pub.min()
        line = "step %2d (%3d wgs): publish first..last %5.2f us (median %5.2f)" % (s_, g, (pub.max() - t0) / 100.0, (np.median(pub) - t0) / 100.0)
        if s_ < 18:
            line += "; folded first..last %5.2f .. %5.2f us after the first publish" % ((fold.min() - t0) / 100.0, (fold.max() - t0) / 100.0)
        if s_ > 0:
            line += "; step length %5.2f us" % ((pub.min() - b[s_ - 1, :grids[s_ - 1], 0].min()) / 100.0)
        print(line)
    late = np.argsort(b[12, :200, 0])[-8:]
    print("latest publishers of step 12:", late, (b[12, late, 0] - b[12, :200, 0].min()) / 100.0)
    # does a group of workgroups (id mod 8 = the XCD under round-robin placement) publish systematically later?
    for s_ in (10, 13, 16):
        d = (b[s_, :200, 0] - b[s_, :200, 0].min()) / 100.0
        print("step %d: mean publish delay by workgroup id mod 8:" % s_, np.round([d[r::8].mean() for r in range(8)], 2))
if os.environ.get("KF_ICP_EXP") == "9":
    # workgroup 5, per wave and step (s_memrealtime, 10 ns ticks): fold done [16..23], pixel phase done [0..7], published [8..15]
    import ctypes as C
    buf = np.zeros(19 * 32, np.uint64)
    ctx.lib.kf_exp_read_icp_slots(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(25 * 512 * 32), C.c_size_t(buf.size))
    b = buf.reshape(19, 32).astype(np.int64)
    for s_ in range(1, 19):
        t0 = b[s_, 16:24].min()
        print("step %2d: fold done %s | solve+pixels done at %s | published at %s (us after the first wave left the fold)" % (
            s_, np.round((b[s_, 16:24] - t0) / 100.0, 2), np.round((b[s_, 0:8] - t0) / 100.0, 2), np.round((b[s_, 8:16] - t0) / 100.0, 2)))
if os.environ.get("KF_ICP_EXP") == "11":
    # workgroup 0, per Gauss-Newton step (s_memrealtime, 10 ns ticks, summed over all steps of all frames since the context was created)
    import ctypes as C
    buf = np.zeros(8, np.uint64)
    ctx.lib.kf_exp_read_icp_slots(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(26 * 512 * 32), C.c_size_t(buf.size))
    n_steps = 18 * (50 + 1)          # 18 applied solves per tracked frame (the 19th is applied after the loop), 50 timed frames + 1 before
    names = ["unpack + Cholesky solve (lane 0)", "hand-off, sin/cos on three lanes, hand-off", "rotation, shake test, T * cur (lane 0)",
             "workgroup barrier", "verdict + copy + barrier", "-", "determinant (lane 64, from the step's entry)"]
    for i, nme in enumerate(names):
        if nme != "-":
            print("  %-48s %6.2f us per step" % (nme, buf[i] / 100.0 / n_steps))
