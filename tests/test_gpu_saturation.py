"""Deferred free-space weights (integrate.hip, k_integrate_pairs<.., DEFER = true>): a quarter brick whose 128 voxels all hold tsdf 1 is not
read or written when all of it is observed as free space again -- the observation is counted in a 16-bit word per quarter and applied to the
weights (w <- fminf(w + k, max_weight)) when anything else writes there, and by kf_download_volume.  Once every weight has reached max_weight
the quarter is saturated and ANY free-space observation of it is the identity.  max_weight 3 reaches saturation after three frames,
max_weight 128 stays in the pending-count state for the whole test.  Everything against the CPU oracle, bit for bit: update counts per frame
and the tsdf / weight planes, from the first frame, through view changes that put surface into deferred space, through download / upload /
reset, with colour frames (which flush and drop the words) in between, and against the same frames with deferral switched off."""
import numpy as np
import pytest

import oracle_lib as O
from hybkinectfu_amd import lib as K
from hybkinectfu_amd import scene as S

pytestmark = pytest.mark.gpu

P = S.STOCK
CAM = (160, 120, 79.5, 59.5, 131.25, 131.25)
CAM_QVGA = (320, 240, 159.5, 119.5, 262.5, 262.5)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class Pair:
    """the same frames through the library and the oracle"""

    def __init__(self, res, size, color=False, cam=CAM, maxw=3.0, defer=1):      # (volumes below 768^3 do not defer unless asked to)
        self.res, self.size, self.color, self.cam, self.maxw = res, size, color, cam, maxw
        self.ocam = O.Cam.make(*cam)
        self.ovol = O.OVolume(res, size, maxw)
        self.ctx = K.Context(K.camera(*cam), res, size, maxw, levels=3, has_color=color)
        if defer is not None:
            self.ctx.set_defer(defer)
        self.rng = np.random.default_rng(3)

    def fuse(self, k, trunc=0.1, max_dist=4.0, color=False, holes=0):
        pose = S.trajectory_pose(k, self.size).astype(np.float32)
        mm = S.render_depth_mm(pose, self.cam, self.size)
        if holes:                                       # sensor drop-outs: waves around them are partial, whatever the quarter's state
            mm = mm.copy()
            mm.reshape(-1)[np.random.default_rng(k).integers(0, mm.size, holes)] = 0
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
        assert st["updated_last"] == n_o, (k, st["updated_last"], n_o)
        self.queued = st["bricks_active"]

    def same_volume(self):
        t, w = self.ctx.download_volume()
        return np.array_equal(bits(t), bits(self.ovol.tsdf)) and np.array_equal(bits(w), bits(self.ovol.weight))

    def close(self):
        self.ctx.close()


@pytest.mark.parametrize("res,maxw", [(64, 3.0), (256, 3.0), (64, 128.0), (256, 128.0), (64, 2.5), (64, 1.0)])
def test_deferred_free_space_is_skipped_without_changing_a_bit(res, maxw):
    p = Pair(res, 3.0, cam=CAM_QVGA if res >= 256 else CAM, maxw=maxw)     # 256^3: bricks ~16 px wide, as at 1024^3 / VGA
    queued = []
    for k in range(10):                                # frame 1 on: whole free-space quarters are counted, not written
        p.fuse(k)
        queued.append(p.queued)
        if k in (0, 1, 2, 5):
            assert p.same_volume(), k                  # (the download applies the pending counts on the fly and leaves the state alone)
    assert p.same_volume()
    print("bricks queued per frame:", queued)
    if res >= 256:                                      # bricks small enough on screen (a few 8- or 16-pixel tiles) to be free space throughout
        assert queued[9] < 0.95 * queued[0]             # whole free-space bricks are retired by the cull (counted, not queued)
    free = int(((p.ovol.weight == min(maxw, 10.0)) & (p.ovol.tsdf == 1.0)).sum())
    assert free > 0.02 * res ** 3                       # the regime is really there: free space observed by all ten frames
    for k in (40, 41, 75, 76, 77, 20):                  # other views: surface bands and frustum edges cut into the deferred space
        p.fuse(k)
        assert p.same_volume(), k
    p.fuse(21, trunc=0.25)                              # a wider band turns free space into band voxels: pending counts are applied first
    p.fuse(22, trunc=0.1, max_dist=1.2)
    assert p.same_volume()
    for k in (23, 24, 25):
        p.fuse(k, holes=40)                             # drop-outs: partial waves over deferred quarters
    assert p.same_volume()
    p.close()


def test_deferral_on_equals_deferral_off():
    a, b = Pair(64, 3.0, maxw=128.0, defer=1), Pair(64, 3.0, maxw=128.0, defer=0)
    for k in list(range(6)) + [40, 41, 7]:
        a.fuse(k); b.fuse(k)
    ta, wa = a.ctx.download_volume()
    tb, wb = b.ctx.download_volume()
    assert np.array_equal(bits(ta), bits(tb)) and np.array_equal(bits(wa), bits(wb))
    b.ctx.set_defer(1); a.ctx.set_defer(0)              # switched in mid-stream: the plain kernel first applies what is pending
    for k in (8, 9, 42, 10):
        a.fuse(k); b.fuse(k)
    assert a.same_volume() and b.same_volume()
    a.close(); b.close()


@pytest.mark.parametrize("maxw", [3.0, 128.0])
def test_deferred_state_survives_download_upload_and_reset(maxw):
    p = Pair(64, 3.0, maxw=maxw)
    for k in range(7):
        p.fuse(k)
    t, w = p.ctx.download_volume()
    q = Pair(64, 3.0, maxw=maxw)
    q.ctx.upload_volume(t, w)                           # the flags are rebuilt and every deferred-weight word dropped
    q.ovol.tsdf[...] = t; q.ovol.weight[...] = w
    for k in range(7, 14):
        p.fuse(k); q.fuse(k)
    assert p.same_volume() and q.same_volume()
    # an upload that covers part of a brick's layers: pending counts of the other layers must have been applied, not lost
    t2, w2 = p.ctx.download_volume(20, 30)
    p.ctx.upload_volume(t2, w2, z0=20)
    for k in range(14, 18):
        p.fuse(k)
    assert p.same_volume()
    p.ctx.reset_volume()
    p.ovol.vox[...] = 0
    for k in range(30, 38):
        p.fuse(k)
    assert p.same_volume()
    p.close(); q.close()


@pytest.mark.parametrize("maxw", [3.0, 128.0])
def test_colour_frames_between_deferred_frames(maxw):
    p = Pair(64, 3.0, color=True, maxw=maxw)
    for k in range(6):
        p.fuse(k)
    p.fuse(6, color=True)                               # the colour kernel blends with the weight: pending counts are applied before it runs
    p.fuse(7, color=True)
    for k in range(8, 14):
        p.fuse(k)
    assert p.same_volume()
    t, w, c = p.ctx.download_volume(color=True)
    seen = p.ovol.weight > 0
    assert np.array_equal(c[seen], p.ovol.color[seen])
    p.close()
