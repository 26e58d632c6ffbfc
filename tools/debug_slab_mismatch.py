#!/usr/bin/env python3
"""Diagnostic: whole-volume context vs N z-slab contexts in ONE process on a trajectory that dollies into the scene / turns; after every frame the
slab raycast candidates are merged as SlabPipeline merges them (first crossing wins) and compared with the whole-volume maps; at the first mismatch
the offending pixels are printed with every slab's crossing parameter, the oracle arbitrates, and the reference's march is replayed sample by sample.
usage: debug_slab_mismatch.py [ranks] [res] [frames] [dolly] [yaw]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, pipeline as PL, scene as S
P = S.STOCK
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dolly = float(sys.argv[4]) if len(sys.argv) > 4 else 0.4
yaw_deg = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
size, cam = 3.0, S.vga_camera()
kcam = K.camera(*cam)
inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
halo = PL.slab_halo_layers(res, size, inc)
ranges = PL.slab_ranges(res, world)
whole = K.Context(kcam, res, size, P["volume_max_weight"], levels=3)
slabs = [K.Context(kcam, res, size, P["volume_max_weight"], levels=3, slab=r, halo=halo) for r in ranges]
dev = torch.device("cuda", 0)
bufs = [(torch.empty((cam[1], cam[0]), dtype=torch.float32, device=dev), torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev),
         torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev)) for _ in slabs]


def pose_of(k):
    p = S.trajectory_pose(k, size); f = k / 31.0; yaw = np.radians(yaw_deg) * f
    turn = np.array([[np.cos(yaw), 0, np.sin(yaw), 0], [0, 1, 0, 0], [-np.sin(yaw), 0, np.cos(yaw), dolly * f], [0, 0, 0, 1.0]])
    return p @ turn


def explain(k, pose_w, wv, mv, bad):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib as O
    f32 = np.float32
    tw, ww = whole.download_volume()
    ovol = O.OVolume(res, size, P["volume_max_weight"]); ovol.tsdf[...] = tw; ovol.weight[...] = ww
    ocam = O.Cam.make(*cam)
    ov, on, _, osteps = O.raycast(ovol, False, pose_w, inc, ocam, P["depth_trunc_min"], P["depth_trunc_max"], want_steps=True)
    print("  oracle == whole-volume kernel: %s; oracle == merged slabs: %s" % (np.array_equal(ov.view(np.uint32), wv.view(np.uint32)), np.array_equal(ov.view(np.uint32), mv.view(np.uint32))))
    y, x = bad[0]
    print("  oracle at the pixel: v %s" % ov[y, x])
    # the ray exactly as raycastKernel :136-150 forms it (fp32 operation by operation; normalize: reciprocal in double, narrowed)
    vx = f32(f32(f32(1.0) * f32(f32(x) - f32(cam[2]))) / f32(cam[4])); vy = f32(f32(f32(1.0) * f32(f32(y) - f32(cam[3]))) / f32(cam[5])); vz = f32(1.0)
    ln = f32(np.sqrt(f32(f32(f32(vx * vx) + f32(vy * vy)) + f32(vz * vz))))
    r = f32(1.0 / float(ln))
    cd = np.array([f32(vx * r), f32(vy * r), f32(vz * r)], np.float32)
    T = pose_w.astype(np.float32)
    d = np.array([f32(f32(f32(T[i, 0] * cd[0]) + f32(T[i, 1] * cd[1])) + f32(T[i, 2] * cd[2])) for i in range(3)], np.float32)
    o = T[:3, 3].copy()
    tmin = max(max(f32(f32((f32(0.0) if d[i] > 0 else f32(size)) - o[i]) / d[i]) for i in range(3)), f32(f32(P["depth_trunc_min"]) / cd[2]))
    t = f32(tmin); last = f32(0)
    print("  oracle took %d samples on this ray (t of the last one %.7f); replay: dir %s tmin %.7f" % (osteps[y, x], float(t) + (int(osteps[y, x]) - 1) * inc, d, t))
    for step in range(120):
        pos = np.array([f32(o[i] + f32(d[i] * t)) for i in range(3)], np.float32)
        g = np.clip(np.array([int(f32(f32(pos[i] * f32(res)) / f32(size))) for i in range(3)]), 0, res - 1)
        sdf = tw[g[2], g[1], g[0]]; wgt = ww[g[2], g[1], g[0]]
        mark = "  <-- last > 0 and cur < 0" if (last > 0 and sdf < 0) else ""
        if t > 1.2 or mark:
            print("    step %3d t %.7f voxel (x %d, y %d, z %d) tsdf %+.5f weight %g%s" % (step, t, g[0], g[1], g[2], sdf, wgt, mark))
        if mark:
            lp = np.array([f32(o[i] + f32(d[i] * f32(t - f32(inc)))) for i in range(3)], np.float32)
            for (pp, nm) in ((pos, "cur"), (lp, "last")):
                okk, val = O.interpolate_sdf(ovol, pp)
                print("       interpolateSDF(%s) -> %s %s" % (nm, okk, val))
        last = sdf; t = f32(t + f32(inc))
        if t > 2.1:
            break


for c in [whole] + slabs:
    c.set_pose(S.pose0(size))
for k in range(n):
    mm = S.render_depth_mm(pose_of(k), cam, size)
    for c in [whole] + slabs:
        c.upload_depth_mm(mm)
        c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        c.integrate(None, P["integrate_sdf_trunc"], P["integrate_depth_trunc"])
    ok_w, pose_w, _, _ = whole.track_result()
    for c in slabs:
        ok_s, pose_s, _, _ = c.track_result()
        assert np.array_equal(pose_s.view(np.uint32), pose_w.view(np.uint32)), "pose differs at frame %d" % k
    for c in slabs:                                           # volumes first: every stored layer of every slab == whole
        z0, z1 = c.stored
        tw_, ww_ = whole.download_volume(z0, z1); t_, w_ = c.download_volume(z0, z1)
        if not (np.array_equal(t_.view(np.uint32), tw_.view(np.uint32)) and np.array_equal(w_, ww_)):
            bad = np.argwhere((t_.view(np.uint32) != tw_.view(np.uint32)) | (w_ != ww_))
            print("frame %d: VOLUME of slab %s (stored %s) differs from the whole volume at %d voxels, first (z, y, x) = %s" % (k, c.owned, c.stored, len(bad), (bad[0] + [z0, 0, 0]).tolist()))
            sys.exit(1)
    whole.raycast(None, inc, P["depth_trunc_min"], P["depth_trunc_max"])
    tas = []
    for c in slabs:
        ta = torch.empty((cam[1], cam[0]), dtype=torch.int64, device=dev)
        c.raycast_slab_cross(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta.data_ptr()); c.sync()
        tas.append(ta)
    ta_min = torch.stack(tas).min(dim=0).values.contiguous()
    acc = torch.zeros((cam[1], cam[0], 4), dtype=torch.int32, device=dev)
    for c in slabs:
        cand = torch.empty((cam[1], cam[0], 4), dtype=torch.float32, device=dev)
        c.slab_ray_normals(None, inc, P["depth_trunc_min"], P["depth_trunc_max"], ta_min.data_ptr(), cand.data_ptr()); c.sync()
        acc += cand.view(torch.int32)
    rays = acc.view(torch.float32).contiguous()
    slabs[0].set_model_maps_rays(None, ta_min.data_ptr(), rays.data_ptr()); slabs[0].sync()
    merged = [torch.from_numpy(slabs[0].download_map(K.MAP_MODEL_VERTICES)).to(dev), torch.from_numpy(slabs[0].download_map(K.MAP_MODEL_NORMALS)).to(dev)]
    for b_, ta in zip(bufs, tas):                              # (for the print-out: the crossing parameters)
        b_[0].copy_((ta >> 32).to(torch.int32).view(torch.float32))
    wv, wn = whole.download_map(K.MAP_MODEL_VERTICES), whole.download_map(K.MAP_MODEL_NORMALS)
    mv, mn = merged[0].cpu().numpy(), merged[1].cpu().numpy()
    same = np.array_equal(mv.view(np.uint32), wv.view(np.uint32)) and np.array_equal(mn.view(np.uint32), wn.view(np.uint32))
    print("frame %2d: camera z %.3f, %d model pixels, merged slabs %s whole" % (k, pose_w[2, 3], int((wv[..., 3] != 0).sum()), "==" if same else "!="), flush=True)
    if not same:
        bad = np.argwhere((mv.view(np.uint32) != wv.view(np.uint32)).any(axis=2) | (mn.view(np.uint32) != wn.view(np.uint32)).any(axis=2))
        print("  %d pixels differ; first ones:" % len(bad))
        tcpu = [b[0].cpu().numpy() for b in bufs]
        for (y, x) in bad[:6]:
            print("  pixel (x %d, y %d): whole v %s n %s | merged v %s n %s | per-slab t_cross %s | per-slab v.w %s" % (
                x, y, wv[y, x], wn[y, x][:3], mv[y, x], mn[y, x][:3], [float(tt[y, x]) for tt in tcpu], [float(b[1][y, x, 3]) for b in bufs]))
            print("     whole-volume vertex z = layer %.2f; slab boundaries %s" % (wv[y, x][2] * res / size, ranges))
        explain(k, pose_w, wv, mv, bad)
        sys.exit(1)
    for c in slabs:
        c.set_model_maps_device(merged[0].data_ptr(), merged[1].data_ptr()); c.sync()
print("no mismatch in %d frames" % n)
