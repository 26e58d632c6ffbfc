#!/usr/bin/env python3
"""Randomised parity sweep (GPU): small random configurations -- resolution (any multiple of 8), volume size, camera size and
intrinsics, truncation, poses off the scripted trajectory, depth holes -- through integrate / raycast / marching cubes and the
slab split, each against the CPU oracle bit for bit.  Not part of the test-suite (minutes of oracle time); run after kernel changes:
    python tools/fuzz_parity.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from hybkinectfu_amd import lib as K, scene as S
P = S.STOCK
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
fails = 0
for case in range(n_cases):
    res = int(rng.choice([24, 32, 40, 56, 64, 72, 96, 104, 128]))
    size = float(rng.choice([1.5, 2.0, 3.0, 4.0]))
    cols, rows = int(rng.integers(40, 200)), int(rng.integers(30, 150))
    f = float(rng.uniform(0.6, 1.2) * cols)
    cam = (cols, rows, cols / 2 - 0.5 + float(rng.uniform(-3, 3)), rows / 2 - 0.5 + float(rng.uniform(-3, 3)), f, f * float(rng.uniform(0.9, 1.1)))
    trunc = float(rng.uniform(2.5, 7.0)) * size / res
    max_dist = float(rng.uniform(0.5, 1.0)) * size
    ocam, kcam = O.Cam.make(*cam), K.camera(*cam)
    ovol = O.OVolume(res, size, 128.0)
    # every third case: the colour path (use_color, with or without the angle weight).  The reference projects into the colour image with its
    # literal 525 / 320 / 240 intrinsics (integrateVolume.cu:56-57), so the colour camera is VGA-sized whatever the depth camera is.
    color = case % 3 == 2
    angled = bool(rng.integers(0, 2))
    rcam = (640, 480, 319.5, 239.5, 525.0, 525.0)
    orcam, krcam = O.Cam.make(*rcam), K.camera(*rcam)
    rgb = rng.integers(0, 256, (480, 640, 3)).astype(np.uint8) if color else None
    ctx = K.Context(kcam, res, size, 128.0, levels=1, max_triangles=600000, has_color=color, rgb_cam=krcam if color else None)
    if color:
        ctx.upload_rgb(rgb)
    ok = True
    pose = None
    for k in range(int(rng.integers(1, 4))):
        pose = S.trajectory_pose(int(rng.integers(0, 100)), size).astype(np.float32)
        pose[:3, 3] += rng.uniform(-0.05, 0.05, 3).astype(np.float32) * size
        mm = S.render_depth_mm(pose, cam, size)
        hole = rng.integers(0, min(rows, cols) // 2, 4)
        mm[hole[0]:hole[0] + hole[1] // 2, hole[2]:hole[2] + hole[3] // 2] = 0
        tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
        fl = O.bilateral(tr, 2.0, 0.03)
        n = O.vertices_to_normals(O.depth_to_vertices(fl, ocam))
        n_o = O.integrate(ovol, tr, n, rgb, color, color and angled, pose, trunc, max_dist, ocam, orcam if color else ocam)
        ctx.upload_depth_mm(mm); ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], 2.0, 0.03)
        if color:                                       # identical normals on both sides: the angle weight reads them (the bilateral differs in last bits)
            ctx.upload_map(K.MAP_NEW_NORMALS, 0, n)
        ctx.integrate(pose, trunc, max_dist, has_color=color, angle_weight=color and angled)
        ok = ok and ctx.stats()["updated_last"] == n_o
    if color:
        t, w, cvol = ctx.download_volume(color=True)
        seen = ovol.weight > 0
        ok = ok and np.array_equal(cvol[seen], ovol.color[seen])
    else:
        t, w = ctx.download_volume()
    ok = ok and np.array_equal(bits(t), bits(ovol.tsdf)) and np.array_equal(bits(w), bits(ovol.weight))
    inc = float(rng.uniform(0.4, 1.5)) * trunc
    ov, on, orgb = O.raycast(ovol, color, pose, inc, ocam, 0.3, 4.0)
    ctx.raycast(pose, inc, 0.3, 4.0, has_color=color)
    ok_r = np.array_equal(bits(ctx.download_map(K.MAP_MODEL_VERTICES)), bits(ov)) and np.array_equal(bits(ctx.download_map(K.MAP_MODEL_NORMALS)), bits(on))
    if color:
        ok_r = ok_r and np.array_equal(ctx.download_map(K.MAP_RAYCAST_RGB), orgb)
    thr = 300 * size / res
    ot = O.marching_cubes(ovol, False, thr, 600000)
    ctx.marching_cubes(thr)
    gt = ctx.triangles()
    ok_m = len(gt) == len(ot) and np.array_equal(gt["v"]["pos"].view(np.uint32), ot["v"]["pos"].view(np.uint32))
    # two z-slabs fed with the whole volume's planes must give the same triangles (slab-major concatenation)
    nb = res // 8
    ok_s = True
    if nb >= 2:
        cut = (nb // 2) * 8
        halo = 8 * int(np.ceil((np.ceil(inc * res / size) + 2) / 8))
        parts = []
        for z0, z1 in ((0, cut), (cut, res)):
            s = K.Context(kcam, res, size, 128.0, levels=1, max_triangles=600000, slab=(z0, z1), halo=halo)
            a, b = s.stored
            s.upload_volume(ovol.tsdf[a:b], ovol.weight[a:b], z0=a)
            s.marching_cubes(thr)
            parts.append(s.triangles()); s.close()
        cat = np.concatenate(parts)
        ok_s = len(cat) == len(ot) and np.array_equal(cat["v"]["pos"].view(np.uint32), ot["v"]["pos"].view(np.uint32))
    ctx.close()
    good = ok and ok_r and ok_m and ok_s
    fails += 0 if good else 1
    print("case %2d: res %3d size %.1f cam %dx%d trunc %.3f inc %.3f%s -> integrate %s raycast %s (%d hits) mcubes %s (%d tris) slabs %s" % (
        case, res, size, cols, rows, trunc, inc, (" colour%s" % (" + angle weight" if angled else "")) if color else "", "ok" if ok else "MISMATCH", "ok" if ok_r else "MISMATCH", int((ov[..., 3] != 0).sum()),
        "ok" if ok_m else "MISMATCH", len(ot), "ok" if ok_s else "MISMATCH"), flush=True)
print("FUZZ %s: %d of %d cases failed" % ("FAILED" if fails else "OK", fails, n_cases))
sys.exit(1 if fails else 0)
