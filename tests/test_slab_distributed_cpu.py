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


def _exchange_worker(rank, world, port, out_dir):
    """SlabExchange (the sequence SlabPipeline runs per frame) over gloo, with the torch restatements of the two kernels."""
    import slab_cpu_ops as ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols = 30, 44
    model = {}
    ex = PL.SlabExchange(rows, cols, torch.device("cpu"), dist, mask=ops.mask,
                         unpack=lambda cand: model.update(zip("vn", ops.unpack(cand))))
    ok = True
    for frame in range(4):
        t, cand, want_v, want_n = ops.synthetic_candidates(rows, cols, rank, world, seed=7 + frame)
        ex.t.copy_(t); ex.cand.copy_(cand)
        calls = []
        ex.merge(lambda: calls.append(1))
        ok = ok and calls == [1]
        ok = ok and torch.equal(model["v"].view(torch.int32), want_v.view(torch.int32))
        ok = ok and torch.equal(model["n"].view(torch.int32), want_n.view(torch.int32))
        # the same rule through the unpacked reference merge
        v, n = ops.unpack(cand)                                  # this rank's own candidates as maps
        mv, mn = PL.merge_candidates(t, v, n, lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN), lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM))
        ok = ok and torch.equal(mv.view(torch.int32), want_v.view(torch.int32)) and torch.equal(mn.view(torch.int32), want_n.view(torch.int32))
    open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("ok" if ok else "mismatch")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_sequence_gloo(tmp_path, world):
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "ok"


def test_bench_launcher_starts_ranks_itself():
    """`python bench.py --gpus 2` with NO WORLD_SIZE/RANK in the environment: the parent (which loads neither torch.cuda nor
    the library) must start two rank processes itself; they rendezvous over gloo, run the pipeline's collective sequence and
    rank 0 prints exactly one JSON line reporting the world size it saw."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--collective-selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert lines[0]["world_size"] == 2 and lines[0]["n_gpus"] == 2 and lines[0]["ok"] is True


def test_bench_refuses_mismatched_world():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--collective-selftest"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_parent_is_gpu_free():
    """The launching parent must not have loaded the HIP library or touched torch.cuda before it starts the ranks."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    head = src[:src.index("def main():")]
    body = src[src.index("def main():"):]
    launch_at = body.index("launch_ranks(args.gpus")
    assert "import torch" not in body[:launch_at] and "lib as K" not in body[:launch_at]
    # module level: only numpy-side helpers are imported
    top = head[:head.index("def workload")]
    assert "import torch" not in top and "lib as K" not in top


def test_halo_follows_ray_increment():
    # C4: inc 0.035 m = 5.97 voxels -> 6 + 2 = 8 layers; 2048^3 @ 4 m: 17.9 voxels -> 20 -> 24 layers (16 would lose crossings)
    assert PL.slab_halo_layers(1024, 6.0, 0.035) == 8
    assert PL.slab_halo_layers(2048, 4.0, 0.035) == 24
    assert PL.slab_halo_layers(512, 4.0, 0.035) == 8
    assert PL.slab_halo_layers(2048, 8.0, 0.035) == 16


def test_balanced_slab_ranges_minimise_the_busiest_rank():
    """pipeline.slab_ranges with per-layer work: contiguous brick-aligned cover, every rank at least one brick layer, and the busiest rank
    (own + halo layers) never busier than under equal thickness -- the dynamic programme is exact, checked against brute force on a small case."""
    import itertools
    import numpy as np
    from hybkinectfu_amd import pipeline as PL
    rng = np.random.default_rng(3)
    for res, world, halo in ((1024, 4, 8), (1024, 8, 8), (512, 2, 8), (2048, 8, 24), (64, 8, 0), (128, 3, 16)):
        nb = res // 8
        z = (np.arange(nb) + 0.5) / nb
        work = np.where(z < 0.75, (z + 0.05) ** 2, 0.0) * rng.uniform(0.8, 1.2, nb)
        hb = (halo + 7) // 8

        def busiest(ranges):
            return max(work[max(0, a // 8 - hb):min(nb, b // 8 + hb)].sum() for a, b in ranges)
        bal, eq = PL.slab_ranges(res, world, work.tolist(), halo=halo), PL.slab_ranges(res, world)
        for ranges in (bal, eq):
            assert ranges[0][0] == 0 and ranges[-1][1] == res and all(a % 8 == 0 and b % 8 == 0 and b > a for a, b in ranges)
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        assert busiest(bal) <= busiest(eq) * (1 + 1e-9)
    # brute force on 12 brick layers, 4 ranks
    res, world = 96, 4
    work = rng.uniform(0.0, 1.0, 12); work[8:] = 0.0
    bal = PL.slab_ranges(res, world, work.tolist(), halo=8)
    best = min(max(work[max(0, c[i] - 1):min(12, c[i + 1] + 1)].sum() for i in range(world))
               for cuts in itertools.combinations(range(1, 12), world - 1) for c in [(0,) + cuts + (12,)])
    got = max(work[max(0, a // 8 - 1):min(12, b // 8 + 1)].sum() for a, b in bal)
    assert got <= best * (1 + 1e-6)
    with pytest.raises(ValueError):
        PL.slab_ranges(64, 9)
