"""CPU restatements of the device launches of the z-slab raycast merge (csrc/raycast.hip: k_slab_rays_unpack; the map form's k_slab_mask) and a
synthetic generator of per-slab crossings.  Test infrastructure: the CPU-only world-2 tests run
pipeline.SlabExchange -- the collective sequence SlabPipeline issues on the GPU -- with these in place of the kernels;
tests/test_gpu_slabs.py checks on the GPU that the kernels produce the same bits as these functions."""
import numpy as np
import torch

CAM = (64, 48, 31.5, 23.5, 52.5, 52.5)           # cols, rows, cx, cy, fx, fy of the synthetic frames
POSE = np.array([[0.9987503, -0.0499792, 0.0, 1.25], [0.0499792, 0.9987503, 0.0, 1.5], [0.0, 0.0, 1.0, 0.125], [0.0, 0.0, 0.0, 1.0]], np.float32)


def mask(t, tmin, cand):
    """k_slab_rays_mask: this rank's candidate survives where it IS the first crossing (t == tmin, finite); losers contribute zero
    bits.  A candidate is (ray parameter of the vertex, normal xyz), 4 floats per pixel; in place."""
    win = ((t == tmin) & torch.isfinite(t)).unsqueeze(-1).to(torch.int32)
    cand.view(torch.int32).mul_(win)


def pixel_rays(pose, cam, rows, cols):
    """rc_pixel_ray (raycastKernel raycastingVolume.cu:136-150) operation for operation in fp32: origin [3], direction [rows, cols, 3]."""
    f = np.float32
    T = np.asarray(pose, f)
    _, _, cx, cy, fx, fy = cam
    vx = np.broadcast_to(((np.arange(cols, dtype=f) - f(cx)) * f(1.0) / f(fx))[None, :], (rows, cols))
    vy = np.broadcast_to(((np.arange(rows, dtype=f) - f(cy)) * f(1.0) / f(fy))[:, None], (rows, cols))
    vz = np.ones((rows, cols), f)
    ln = np.sqrt((vx * vx + vy * vy) + vz * vz)
    r = (1.0 / ln.astype(np.float64)).astype(f)              # cuda_declar.h:89-94: reciprocal in double, narrowed
    d = [vx * r, vy * r, vz * r]
    out = np.empty((rows, cols, 3), f)
    for k in range(3):
        w = ((T[k, 0] * d[0] + T[k, 1] * d[1]) + T[k, 2] * d[2]) + T[k, 3] * f(0.0)
        out[..., k] = np.where(w == 0, f(1e-15), w)
    return T[:3, 3].copy(), out


def pack_ta(t, alpha):
    """(bits of the crossing's ray parameter) << 32 | bits of the vertex's ray parameter, as int64 -- what kf_raycast_volume_slab_cross writes"""
    hi = t.contiguous().view(torch.int32).to(torch.int64) << 32
    lo = alpha.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return hi | lo


def ta_alpha(ta):
    return torch.from_numpy((ta.numpy() & 0xFFFFFFFF).astype(np.uint32).view(np.float32))


def unpack(ta, cand, pose=POSE, cam=CAM):
    """k_slab_rays_unpack: where the vertex's owner found a gradient (cand = its three words, some bit set -- a unit vector): vertex = origin + direction * alpha,
    w = 1, normal = cand with w = 0; zeros elsewhere."""
    c = cand.numpy()
    rows, cols = c.shape[:2]
    org, dirs = pixel_rays(pose, cam, rows, cols)
    valid = (c.view(np.uint32) != 0).any(axis=-1)
    alpha = ta_alpha(ta).numpy()
    v = np.zeros((rows, cols, 4), np.float32)
    n = np.zeros_like(v)
    vt = org[None, None, :] + dirs * alpha[..., None]
    v[..., 0:3] = np.where(valid[..., None], vt, np.float32(0.0))
    v[..., 3] = valid.astype(np.float32)
    n[..., 0:3] = np.where(valid[..., None], c, np.float32(0.0))
    return torch.from_numpy(v), torch.from_numpy(n)


def synthetic_crossings(rows, cols, rank, world, seed):
    """What `world` slabs would report for one frame, generated identically on every rank from `seed`.  Per pixel: the slab that meets the first crossing
    (or none), its parameter t and the vertex's parameter alpha -- sometimes far beyond t, the extrapolation that sends the vertex into ANOTHER slab;
    alpha 0 where the reference gives up at the crossing (the empty pixel must still win) --, later slabs reporting a losing crossing further along the
    ray, the slab that owns the vertex (independent of the one that met the crossing) and the normal it finds (none where the gradient fails).
    Returns this rank's words, its `normals(ta_min, cand)` step (which also checks that the MIN all-reduce delivered the winners) and the merged model
    maps every rank must end up with."""
    g = torch.Generator().manual_seed(seed)
    crosser = torch.randint(0, world + 1, (rows, cols), generator=g)            # == world: no crossing anywhere
    t_true = torch.rand((rows, cols), generator=g) * 3 + 0.3
    alpha = t_true - 0.01 * torch.rand((rows, cols), generator=g)              # usually the vertex lies a little before the crossing's sample ...
    far = torch.rand((rows, cols), generator=g) < 0.1
    alpha = torch.where(far, t_true + 2.0 * torch.rand((rows, cols), generator=g), alpha)      # ... sometimes far beyond it
    gave_up = torch.rand((rows, cols), generator=g) < 0.15
    alpha = torch.where(gave_up, torch.zeros_like(alpha), alpha)
    v_owner = torch.randint(0, world, (rows, cols), generator=g)
    n_true = torch.randn((rows, cols, 3), generator=g)
    n_true[0, 0] = torch.tensor([-0.0, 1.0, -0.0])                              # signed zeros must survive the integer sum
    no_grad = torch.rand((rows, cols), generator=g) < 0.1
    inf = torch.full_like(t_true, float("inf"))
    mine = crosser == rank
    later = (crosser < rank) & (crosser < world)
    t = torch.where(mine, t_true, torch.where(later, t_true + 0.5, inf))
    a = torch.where(mine, alpha, torch.where(later, torch.full_like(alpha, 7.0), torch.zeros_like(alpha)))
    ta = pack_ta(t, a)
    has = crosser < world
    want_ta = pack_ta(torch.where(has, t_true, inf), torch.where(has, alpha, torch.zeros_like(alpha)))
    valid = has & ~gave_up & ~no_grad

    def normals(ta_min, cand):
        assert torch.equal(ta_min, want_ta), "the MIN all-reduce did not deliver the first crossings"
        cand.zero_()
        sel = valid & (v_owner == rank)
        cand.copy_(torch.where(sel.unsqueeze(-1), n_true, torch.zeros_like(n_true)))

    full = torch.where(valid.unsqueeze(-1), n_true, torch.zeros_like(n_true))
    want_v, want_n = unpack(want_ta, full.contiguous())
    return ta.contiguous(), normals, want_ta, full.contiguous(), want_v, want_n


# ---- a numpy stand-in for one rank's slab of the volume: what SlabMigrator moves, without a GPU ---------------------------------------------------
class FakeSlab:
    """The volume operations SlabMigrator needs, with kf_resize_slab's semantics, on a numpy array: layers stored before and after a resize keep
    their content, new layers read as zeros (never observed) until imported."""

    def __init__(self, res, owned, halo, xy=4):
        from hybkinectfu_amd import pipeline as PL
        self.PL, self.res, self.halo, self.xy = PL, res, halo, xy
        self.owned = tuple(owned)
        self.stored = PL.stored_range(self.owned, halo, res)
        self.data = np.zeros((2, self.stored[1] - self.stored[0], xy, xy), np.float32)

    @staticmethod
    def truth(z0, z1, xy, frame=0):
        """what layers [z0, z1) hold after `frame` fused frames: a function of (plane, z, y, x) -- any rank that integrates a layer arrives at it"""
        z = np.arange(z0, z1, dtype=np.float32).reshape(1, -1, 1, 1)
        y = np.arange(xy, dtype=np.float32).reshape(1, 1, -1, 1)
        x = np.arange(xy, dtype=np.float32).reshape(1, 1, 1, -1)
        p = np.arange(2, dtype=np.float32).reshape(-1, 1, 1, 1)
        return (np.sin(z * 0.37 + y * 1.3 + x * 2.1 + p) + np.float32(frame) * (z + 1.0)).astype(np.float32)

    def integrate(self, frame):
        self.data[...] = self.truth(self.stored[0], self.stored[1], self.xy, frame)      # own AND halo layers, as the real slab re-integrates both

    def export(self, z0, z1):
        assert self.stored[0] <= z0 < z1 <= self.stored[1]
        return torch.from_numpy(self.data[:, z0 - self.stored[0]:z1 - self.stored[0]].copy())

    def import_(self, z0, z1, t):
        assert self.stored[0] <= z0 < z1 <= self.stored[1]
        self.data[:, z0 - self.stored[0]:z1 - self.stored[0]] = t.numpy()

    def resize(self, z0, z1):
        new_stored = self.PL.stored_range((z0, z1), self.halo, self.res)
        fresh = np.zeros((2, new_stored[1] - new_stored[0], self.xy, self.xy), np.float32)
        lo, hi = max(new_stored[0], self.stored[0]), min(new_stored[1], self.stored[1])
        if hi > lo:
            fresh[:, lo - new_stored[0]:hi - new_stored[0]] = self.data[:, lo - self.stored[0]:hi - self.stored[0]]
        self.data, self.stored, self.owned = fresh, new_stored, (z0, z1)

    def alloc(self, z0, z1):
        return torch.empty((2, z1 - z0, self.xy, self.xy), dtype=torch.float32)
