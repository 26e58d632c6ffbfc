"""CPU-only (gloo, world_size 2): the collective part of the z-slab pipeline -- slab bookkeeping and the raycast candidate merge
(MIN all-reduce of the crossing parameter, integer SUM all-reduce of the winner's maps).  The kernels that produce the
candidates need a GPU; their slab-vs-whole equivalence is covered by tests/test_gpu_slabs.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hybkinectfu_amd import pipeline as PL
from hybkinectfu_amd import scene as S


def test_slab_ranges_cover_volume():
    for res in (64, 128, 512, 1024, 2048):
        for world in (1, 2, 3, 4, 8):
            r = PL.slab_ranges(res, world)
            assert r[0][0] == 0 and r[-1][1] == res
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all((z1 - z0) % 8 == 0 and z1 > z0 for z0, z1 in r)
            sizes = [z1 - z0 for z0, z1 in r]
            assert max(sizes) - min(sizes) <= 8


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1234)                 # identical on both ranks: the "true" per-pixel outcome
    H, W = 24, 32
    owner = torch.randint(0, world + 1, (H, W), generator=g)              # world = no crossing anywhere
    t_true = torch.rand((H, W), generator=g) * 3 + 0.3
    v_true = torch.randn((H, W, 4), generator=g)
    n_true = torch.randn((H, W, 4), generator=g)
    v_true[0, 0] = torch.tensor([-0.0, 1.0, -0.0, 1.0])                   # signed zeros must survive the merge
    failed = torch.rand((H, W), generator=g) < 0.2                         # crossings where the march gave up: zeros win
    v_true[failed] = 0
    n_true[failed] = 0
    # a later slab may also see a (losing) crossing further along the ray
    t_late = t_true + 0.5
    mine = owner == rank
    later = (owner < rank) & (owner < world)
    t = torch.where(mine, t_true, torch.where(later, t_late, torch.full_like(t_true, float("inf"))))
    v = torch.where(mine.unsqueeze(-1), v_true, torch.where(later.unsqueeze(-1), torch.full_like(v_true, 7.0), torch.zeros_like(v_true)))
    n = torch.where(mine.unsqueeze(-1), n_true, torch.where(later.unsqueeze(-1), torch.full_like(n_true, 7.0), torch.zeros_like(n_true)))
    mv, mn = PL.merge_candidates(t.contiguous(), v.contiguous(), n.contiguous(),
                                 lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN), lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM))
    has = owner < world
    exp_v = torch.where(has.unsqueeze(-1), v_true, torch.zeros_like(v_true))
    exp_n = torch.where(has.unsqueeze(-1), n_true, torch.zeros_like(n_true))
    ok = torch.equal(mv.view(torch.int32), exp_v.view(torch.int32)) and torch.equal(mn.view(torch.int32), exp_n.view(torch.int32))
    open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def test_merge_candidates_gloo_world2(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "ok"
