"""Plain-torch restatements of the two device launches of the z-slab raycast merge (csrc/raycast.hip: k_slab_pack,
k_slab_unpack) and a synthetic candidate generator.  Test infrastructure: the CPU-only world-2 tests run
pipeline.SlabExchange -- the collective sequence SlabPipeline issues on the GPU -- with these in place of the kernels;
tests/test_gpu_slabs.py checks on the GPU that the kernels produce the same bits as these functions."""
import torch


def pack(t, tmin, v, n, packed):
    """k_slab_pack: this rank's candidate survives where it IS the first crossing (t == tmin, finite); losers contribute
    zero bits.  Vertex xyz + normal xyz, 6 floats per pixel."""
    win = ((t == tmin) & torch.isfinite(t)).unsqueeze(-1).to(torch.int32)
    packed.view(torch.int32)[..., 0:3] = v.view(torch.int32)[..., 0:3] * win
    packed.view(torch.int32)[..., 3:6] = n.view(torch.int32)[..., 0:3] * win


def unpack(packed):
    """k_slab_unpack: a unit normal is never all-zero, which marks a valid pixel (vertex w = 1, else the zero vertex)."""
    nrm = packed[..., 3:6]
    valid = (nrm != 0).any(dim=-1, keepdim=True)
    v = torch.zeros(packed.shape[:-1] + (4,), dtype=torch.float32)
    n = torch.zeros_like(v)
    v[..., 0:3] = torch.where(valid, packed[..., 0:3], torch.zeros_like(nrm))
    v[..., 3:4] = valid.to(torch.float32)
    n[..., 0:3] = nrm
    return v, n


def synthetic_candidates(rows, cols, rank, world, seed):
    """What `world` slabs would report for one frame, generated identically on every rank from `seed`: per pixel an owner
    slab (or none), the owner's crossing (some of them 'failed': the reference gives up there and leaves zeros -- the zeros must
    still win), later slabs report a losing crossing further along the ray.  Returns this rank's (t, v, n) and the merged
    maps every rank must end up with."""
    g = torch.Generator().manual_seed(seed)
    owner = torch.randint(0, world + 1, (rows, cols), generator=g)              # == world: no crossing anywhere
    t_true = torch.rand((rows, cols), generator=g) * 3 + 0.3
    v_true = torch.randn((rows, cols, 4), generator=g)
    n_true = torch.randn((rows, cols, 4), generator=g)
    v_true[..., 3] = 1.0
    n_true[..., 3] = 0.0
    v_true[0, 0, :3] = torch.tensor([-0.0, 1.0, -0.0])                         # signed zeros must survive the integer sum
    failed = torch.rand((rows, cols), generator=g) < 0.2
    v_true[failed] = 0
    n_true[failed] = 0
    mine = owner == rank
    later = (owner < rank) & (owner < world)
    inf = torch.full_like(t_true, float("inf"))
    t = torch.where(mine, t_true, torch.where(later, t_true + 0.5, inf))
    seven = torch.full_like(v_true, 7.0)
    zero = torch.zeros_like(v_true)
    v = torch.where(mine.unsqueeze(-1), v_true, torch.where(later.unsqueeze(-1), seven, zero))
    n = torch.where(mine.unsqueeze(-1), n_true, torch.where(later.unsqueeze(-1), seven, zero))
    has = (owner < world).unsqueeze(-1)
    return t.contiguous(), v.contiguous(), n.contiguous(), torch.where(has, v_true, zero), torch.where(has, n_true, zero)
