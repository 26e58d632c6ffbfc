#!/usr/bin/env python3
"""How far is the brick queue from the smallest possible one?  Evaluates the reference's update predicate for every voxel of the
C2 / C4 volume with torch on the GPU (float32, approximate at the boundaries -- this is a statistic, not a parity check) and
counts the bricks that hold at least one updating voxel; compares with what the cull queued."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.workload(1, cfg)
cam, P = wl["cam"], S.STOCK
res, size = wl["res"], wl["size"]
pose = S.trajectory_pose(3, size).astype(np.float32)
mm = S.render_depth_mm(pose, cam, size)
ctx = K.Context(K.camera(*cam), res, size, P["volume_max_weight"], levels=3)
ctx.upload_depth_mm(mm)
ctx.preprocess(P["depth_trunc_min"], wl["trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
ctx.integrate(pose, P["integrate_sdf_trunc"], wl["integ_dist"])
st = ctx.stats()
depth = torch.from_numpy(ctx.download_map(K.MAP_TRUNCED_DEPTH)).cuda()
Tinv = torch.from_numpy(np.linalg.inv(pose.astype(np.float64)).astype(np.float32)).cuda()
cols, rows, cx, cy, fx, fy = cam
cell = size / res
trunc, maxd = P["integrate_sdf_trunc"], wl["integ_dist"]
live_bricks = 0; n_upd = 0
hist = torch.zeros(513, dtype=torch.int64, device="cuda")
ax = (torch.arange(res, device="cuda", dtype=torch.float32) + 0.5) * cell
for z0 in range(0, res, 8):
    z = ax[z0:z0 + 8].view(8, 1, 1); y = ax.view(1, res, 1); x = ax.view(1, 1, res)
    pfx = Tinv[0, 0] * x + Tinv[0, 1] * y + Tinv[0, 2] * z + Tinv[0, 3]
    pfy = Tinv[1, 0] * x + Tinv[1, 1] * y + Tinv[1, 2] * z + Tinv[1, 3]
    pfz = Tinv[2, 0] * x + Tinv[2, 1] * y + Tinv[2, 2] * z + Tinv[2, 3]
    ok = pfz > 0
    sx = torch.floor(pfx * fx / pfz + cx + 0.5).long(); sy = torch.floor(pfy * fy / pfz + cy + 0.5).long()
    ok &= (sx >= 1) & (sx < cols - 1) & (sy >= 1) & (sy < rows - 1)
    d = depth[sy.clamp(0, rows - 1), sx.clamp(0, cols - 1)]
    upd = ok & (d != 0) & (d < maxd) & ((d - pfz) > -trunc)
    per = upd.view(8, res // 8, 8, res // 8, 8).sum(dim=(0, 2, 4)).flatten()
    live_bricks += int((per > 0).sum()); n_upd += int(per.sum())
    hist += torch.bincount(per, minlength=513)
h = hist.cpu().numpy()
print("%s: voxels updated %d (library %d); bricks with >= 1 updating voxel %d; queued by the cull %d (%.0f %% above the minimum)" % (
    cfg, n_upd, st["updated_last"], live_bricks, st["bricks_active"], 100.0 * (st["bricks_active"] / live_bricks - 1)))
print("of the bricks that need fusing: %d hold <= 64 updating voxels, %d hold 65-256, %d hold 257-511, %d are full" % (
    h[1:65].sum(), h[65:257].sum(), h[257:512].sum(), h[512]))

# ---- finer than a brick: would a second cull level pay?  4x4x4 octants (64 voxels) of the bricks that need fusing -------------------
live_oct = 0; upd_in_live_oct = 0; oct_total = 0
for z0 in range(0, res, 8):
    z = ax[z0:z0 + 8].view(8, 1, 1); y = ax.view(1, res, 1); x = ax.view(1, 1, res)
    pfx = Tinv[0, 0] * x + Tinv[0, 1] * y + Tinv[0, 2] * z + Tinv[0, 3]
    pfy = Tinv[1, 0] * x + Tinv[1, 1] * y + Tinv[1, 2] * z + Tinv[1, 3]
    pfz = Tinv[2, 0] * x + Tinv[2, 1] * y + Tinv[2, 2] * z + Tinv[2, 3]
    ok = pfz > 0
    sx = torch.floor(pfx * fx / pfz + cx + 0.5).long(); sy = torch.floor(pfy * fy / pfz + cy + 0.5).long()
    ok &= (sx >= 1) & (sx < cols - 1) & (sy >= 1) & (sy < rows - 1)
    d = depth[sy.clamp(0, rows - 1), sx.clamp(0, cols - 1)]
    upd = ok & (d != 0) & (d < maxd) & ((d - pfz) > -trunc)
    o = upd.view(2, 4, res // 4, 4, res // 4, 4).sum(dim=(1, 3, 5))          # [2, res/4, res/4] octant counts
    brick_live = (o.view(2, res // 8, 2, res // 8, 2).sum(dim=(0, 2, 4)) > 0)  # [res/8, res/8]
    bl = brick_live.view(1, res // 8, 1, res // 8, 1).expand(2, res // 8, 2, res // 8, 2).reshape(2, res // 4, res // 4)
    oct_total += int(bl.sum()); live_oct += int(((o > 0) & bl).sum()); upd_in_live_oct += int(o[(o > 0) & bl].sum())
print("octants (4^3) of those bricks: %d of %d hold an updating voxel (%.0f %%); %.1f of 64 voxels update in such an octant" % (
    live_oct, oct_total, 100.0 * live_oct / oct_total, upd_in_live_oct / max(live_oct, 1)))
