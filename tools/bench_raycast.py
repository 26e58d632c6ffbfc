#!/usr/bin/env python3
"""Micro-benchmark of the raycast pass alone on C2 / C4 after fusing a few frames (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
ctx = K.Context(K.camera(*cam), wl["res"], wl["size"], P["volume_max_weight"], levels=3)
for k in range(4):
    pose = S.trajectory_pose(k, wl["size"]).astype(np.float32)
    ctx.upload_depth_mm(S.render_depth_mm(pose, cam, wl["size"]))
    ctx.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
ctx.stage_timers(1 << 4)
for _ in range(30):
    ctx.raycast(pose, 0.035, P["depth_trunc_min"], wl["trunc_max"])
ms, cnt = ctx.read_stage_ms()
v = ctx.download_map(K.MAP_MODEL_VERTICES)
print("%s raycast: %.4f ms per frame, %d hit pixels of %d" % (cfg, ms[4] / cnt[4], int((v[..., 3] != 0).sum()), v.shape[0] * v.shape[1]))
if os.environ.get("KF_RAYCAST_EXP") == "3":
    n = ctx.download_map(K.MAP_MODEL_NORMALS)
    for i, name in enumerate(["prologue ticks", "march ticks", "eval ticks", "loop trips"]):
        a = v[..., i]
        print("%-15s mean %9.0f  p50 %9.0f  p99 %9.0f  max %9.0f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 99), a.max()))
    for i, name in enumerate(["voxel samples", "macro-cell skips", "super-cell skips"]):
        a = n[..., i]
        print("%-15s mean %9.1f  p50 %9.0f  p99 %9.0f  max %9.0f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 99), a.max()))
    # what the slowest lane of each wave spends its trips on
    H, W = v.shape[:2]
    tr = v[..., 3].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    sm = n[..., 0].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    mc = (n[..., 1] + n[..., 2]).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    k = tr.argmax(axis=1); r = np.arange(len(k))
    print("slowest lane per wave: trips %.1f = samples %.1f + macro+super skips %.1f + brick skips %.1f" % (tr[r, k].mean(), sm[r, k].mean(), mc[r, k].mean(), (tr[r, k] - sm[r, k] - mc[r, k]).mean()))
    top = np.argsort(tr[r, k])[-len(k) // 20:]
    print("slowest 5%% of waves:   trips %.1f = samples %.1f + macro+super skips %.1f + brick skips %.1f" % (tr[r, k][top].mean(), sm[r, k][top].mean(), mc[r, k][top].mean(), (tr[r, k] - sm[r, k] - mc[r, k])[top].mean()))
    print("all rays:              trips %.1f = samples %.1f + macro+super skips %.1f + brick skips %.1f" % (tr.mean(), sm.mean(), mc.mean(), (tr - sm - mc).mean()))
    # what a wave's march time is made of: least squares of the per-wave march ticks on its longest lane's trips and samples
    mt = v[..., 1].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)[:, 0]
    A = np.stack([np.ones(len(mt)), tr.max(axis=1), sm.max(axis=1), (sm > 0).sum(axis=1)], axis=1)
    coef = np.linalg.lstsq(A, mt, rcond=None)[0]
    print("march ticks per wave ~ %.0f + %.0f x max trips + %.0f x max samples + %.0f x sampling lanes   (mean %.0f, residual rms %.0f)" % (coef[0], coef[1], coef[2], coef[3], mt.mean(), np.sqrt(((A @ coef - mt) ** 2).mean())))
    slow = np.argsort(mt)[-len(mt) // 50:]
    print("slowest 2%% of waves: march ticks %.0f, max trips %.1f, max samples %.1f, sampling lanes %.1f, eval ticks %.0f" % (mt[slow].mean(), tr.max(axis=1)[slow].mean(), sm.max(axis=1)[slow].mean(), (sm > 0).sum(axis=1)[slow].mean(), v[..., 2].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)[:, 0][slow].mean()))
    # how coherent are the slow waves?  (would finishing a few long rays cooperatively inside their wave help, or are all 64 rays of a slow wave long?)
    mx = tr.max(axis=1)
    slowest = np.argsort(mx)[-len(mx) // 20:]
    for thr in (8, 12):
        act = (tr[slowest] > thr).sum(axis=1)
        print("slowest 5%% of waves: lanes still marching after %2d trips: mean %.1f, median %.0f, <= 8 lanes in %.0f %% of them; their max trips %.1f, mean trips %.1f" % (
            thr, act.mean(), np.median(act), 100.0 * (act <= 8).mean(), mx[slowest].mean(), tr[slowest].mean()))
    wg = mx.reshape(H // 8, W // 8)
    wgmax = wg.reshape(H // 16, 2, W // 32, 4).max(axis=(1, 3))
    print("per-WORKGROUP max trips: mean %.1f p99 %.0f max %.0f; waves of the slowest 5%% of workgroups: mean of their wave-max %.1f" % (
        wgmax.mean(), np.percentile(wgmax, 99), wgmax.max(), wg.reshape(H // 16, 2, W // 32, 4).transpose(0, 2, 1, 3).reshape(-1, 8)[np.argsort(wgmax.reshape(-1))[-len(wgmax.reshape(-1)) // 20:]].mean()))
    # per 8x8 patch (= one wave): the wave runs as long as its slowest lane
    trips = v[..., 3].reshape(v.shape[0] // 8, 8, v.shape[1] // 8, 8).max(axis=(1, 3))
    print("per-wave max trips: mean %.1f  p99 %.0f  max %.0f" % (trips.mean(), np.percentile(trips, 99), trips.max()))
