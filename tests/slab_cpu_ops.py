"""CPU restatements of the two device launches of the z-slab raycast merge (csrc/raycast.hip: k_slab_rays_mask,
k_slab_rays_unpack) and a synthetic candidate generator.  Test infrastructure: the CPU-only world-2 tests run
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


def unpack(cand, pose=POSE, cam=CAM):
    """k_slab_rays_unpack: a unit normal is never all-zero, which marks a valid pixel: vertex = origin + direction * parameter, w = 1
    (else the zero vertex); the normal's w is 0."""
    c = cand.numpy()
    rows, cols = c.shape[:2]
    org, dirs = pixel_rays(pose, cam, rows, cols)
    valid = (c[..., 1:4] != 0).any(axis=-1)
    v = np.zeros((rows, cols, 4), np.float32)
    n = np.zeros_like(v)
    vt = org[None, None, :] + dirs * c[..., 0:1]
    v[..., 0:3] = np.where(valid[..., None], vt, np.float32(0.0))
    v[..., 3] = valid.astype(np.float32)
    n[..., 0:3] = c[..., 1:4]
    return torch.from_numpy(v), torch.from_numpy(n)


def synthetic_candidates(rows, cols, rank, world, seed):
    """What `world` slabs would report for one frame, generated identically on every rank from `seed`: per pixel an owner
    slab (or none), the owner's crossing (some of them 'failed': the reference gives up there and leaves zeros -- the zeros must
    still win), later slabs report a losing crossing further along the ray.  Returns this rank's (t, cand) and the merged model
    maps (vertex, normal) every rank must end up with."""
    g = torch.Generator().manual_seed(seed)
    owner = torch.randint(0, world + 1, (rows, cols), generator=g)              # == world: no crossing anywhere
    t_true = torch.rand((rows, cols), generator=g) * 3 + 0.3
    c_true = torch.randn((rows, cols, 4), generator=g)
    c_true[..., 0] = t_true - 0.01 * torch.rand((rows, cols), generator=g)     # the vertex lies a little before the crossing's sample
    c_true[0, 0, 1:] = torch.tensor([-0.0, 1.0, -0.0])                         # signed zeros must survive the integer sum
    failed = torch.rand((rows, cols), generator=g) < 0.2
    c_true[failed] = 0
    mine = owner == rank
    later = (owner < rank) & (owner < world)
    inf = torch.full_like(t_true, float("inf"))
    t = torch.where(mine, t_true, torch.where(later, t_true + 0.5, inf))
    seven = torch.full_like(c_true, 7.0)
    zero = torch.zeros_like(c_true)
    cand = torch.where(mine.unsqueeze(-1), c_true, torch.where(later.unsqueeze(-1), seven, zero))
    has = (owner < world).unsqueeze(-1)
    want_v, want_n = unpack(torch.where(has, c_true, zero).contiguous())
    return t.contiguous(), cand.contiguous(), want_v, want_n


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
