"""The fusion pass once weights have saturated (integrate.hip, k_integrate_pairs<.., SAT = true>): free space that is (tsdf 1, weight
max_weight) is no longer read or written when it is observed as free space again -- the update would be the identity.  A small
max_weight makes the regime start after three frames instead of 128.  Everything against the CPU oracle, bit for bit: update counts
per frame and the tsdf / weight planes, through saturation, through a view change that puts surface into saturated space, through
download / upload / reset, and with colour frames (the scalar kernel, which only drops the saturation bits) in between."""
import numpy as np
import pytest

import oracle_lib as O
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import scene as S

pytestmark = pytest.mark.gpu

P = S.STOCK
CAM = (160, 120, 79.5, 59.5, 131.25, 131.25)
MAXW = 3.0


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class Pair:
    """the same frames through the library and the oracle"""

    def __init__(self, res, size, color=False, cam=CAM):
        self.res, self.size, self.color, self.cam = res, size, color, cam
        self.ocam = O.Cam.make(*cam)
        self.ovol = O.OVolume(res, size, MAXW)
        self.ctx = K.Context(K.camera(*cam), res, size, MAXW, levels=3, has_color=color)
        self.rng = np.random.default_rng(3)

    def fuse(self, k, trunc=0.1, max_dist=4.0, color=False):
        pose = S.trajectory_pose(k, self.size).astype(np.float32)
        mm = S.render_depth_mm(pose, self.cam, self.size)
        tr = O.trunc_depth(O.depth_mm_to_m(mm), P["depth_trunc_min"], P["depth_trunc_max"])
        n = O.vertices_to_normals(O.depth_to_vertices(O.bilateral(tr, P["filter_sigma_pixel"], P["filter_sigma_depth"]), self.ocam))
        rgb = self.rng.integers(0, 256, (self.cam[1], self.cam[0], 3)).astype(np.uint8) if color else None
        n_o = O.integrate(self.ovol, tr, n, rgb, color, color, pose, trunc, max_dist, self.ocam, self.ocam)
        self.ctx.upload_depth_mm(mm)
        self.ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        if color:
            self.ctx.upload_rgb(rgb)
            self.ctx.upload_map(K.MAP_NEW_NORMALS, 0, n)
        self.ctx.integrate(pose, trunc, max_dist, has_color=color, angle_weight=color)
        st = self.ctx.stats()
        assert st["updated_last"] == n_o, (k, n_o)
        self.queued = st["bricks_active"]

    def same_volume(self):
        t, w = self.ctx.download_volume()
        return np.array_equal(bits(t), bits(self.ovol.tsdf)) and np.array_equal(bits(w), bits(self.ovol.weight))

    def close(self):
        self.ctx.close()


@pytest.mark.parametrize("res", [64, 256])
def test_saturated_free_space_is_skipped_without_changing_a_bit(res):
    p = Pair(res, 3.0, cam=(320, 240, 159.5, 119.5, 262.5, 262.5) if res >= 256 else CAM)     # 256^3: bricks ~16 px wide, as at 1024^3 / VGA
    queued = []
    for k in range(10):                                # frames 3.. run the saturation-aware kernel; frames 4.. find saturated quarters
        p.fuse(k)
        queued.append(p.queued)
    assert p.same_volume()
    print("bricks queued per frame:", queued)
    if res >= 256:                                      # bricks small enough on screen (a few 8- or 16-pixel tiles) to be free space throughout
        assert queued[9] < 0.95 * queued[2]             # whole saturated bricks are retired by the cull (counted, not queued)
    sat = int(((p.ovol.weight == MAXW) & (p.ovol.tsdf == 1.0)).sum())
    assert sat > 0.02 * res ** 3                        # the regime is really there: free space at (1, max_weight)
    for k in (40, 41, 75, 76, 77, 20):                  # other views: surface bands and frustum edges cut into the saturated space
        p.fuse(k)
        assert p.same_volume(), k
    p.fuse(21, trunc=0.25)                              # a wider band turns free space into band voxels: bits must drop
    p.fuse(22, trunc=0.1, max_dist=1.2)
    assert p.same_volume()
    p.close()


def test_saturation_survives_download_upload_and_reset():
    p = Pair(64, 3.0)
    for k in range(7):
        p.fuse(k)
    t, w = p.ctx.download_volume()
    q = Pair(64, 3.0)
    q.ctx.upload_volume(t, w)                           # the flags are rebuilt: no saturation bit, and the plain kernel runs again for 3 frames
    q.ovol.tsdf[...] = t; q.ovol.weight[...] = w
    for k in range(7, 14):
        p.fuse(k); q.fuse(k)
    assert p.same_volume() and q.same_volume()
    p.ctx.reset_volume()
    p.ovol.vox[...] = 0
    for k in range(30, 38):
        p.fuse(k)
    assert p.same_volume()
    p.close(); q.close()


def test_colour_frames_between_saturated_frames():
    p = Pair(64, 3.0, color=True)
    for k in range(6):
        p.fuse(k)
    p.fuse(6, color=True)                               # scalar kernel: writes the (unchanged) voxels and drops the bits of what it touches
    p.fuse(7, color=True)
    for k in range(8, 14):
        p.fuse(k)
    assert p.same_volume()
    p.close()
