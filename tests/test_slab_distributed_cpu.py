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
    """SlabExchange (the sequence SlabPipeline runs per frame) over gloo, with restatements of the device launches."""
    import slab_cpu_ops as ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols = 30, 44
    model = {}
    step = {}
    ex = PL.SlabExchange(rows, cols, torch.device("cpu"), dist, normals=lambda ta, cand: step["normals"](ta, cand),
                         unpack=lambda ta, cand: model.update(zip("vn", ops.unpack(ta, cand))))
    ok = True
    for frame in range(4):
        ta, normals, want_ta, want_cand, want_v, want_n = ops.synthetic_crossings(rows, cols, rank, world, seed=7 + frame)
        step["normals"] = normals
        ex.ta.copy_(ta)
        calls = []
        ex.merge(lambda: calls.append(1))
        ok = ok and calls == [1]
        ok = ok and torch.equal(ex.ta, want_ta) and torch.equal(ex.cand.view(torch.int32), want_cand.view(torch.int32))
        ok = ok and torch.equal(model["v"].view(torch.int32), want_v.view(torch.int32))
        ok = ok and torch.equal(model["n"].view(torch.int32), want_n.view(torch.int32))
        ok = ok and int((model["v"][..., 3] == 1).sum()) > 300
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


def test_a_migration_must_pay_for_itself():
    """pipeline.migration_pays: moving the boundaries costs the wire time of the layers the busiest receiver waits for and gains the busiest rank's
    work reduction on the part of a frame that shards, over the frames until the next decision.  The 1024^3 oscillation seen in a 4-rank rehearsal
    (88 layers of 8 MiB each for a 20 % gain over 20 frames of 0.35 ms) must be declined; the same plan over a long horizon, or on a volume whose
    layers are small, goes through; a plan that gains nothing never pays."""
    from hybkinectfu_amd import pipeline as PL
    res, halo = 1024, 8
    a = [(0, 320), (320, 448), (448, 592), (592, 1024)]
    b = [(0, 408), (408, 544), (544, 672), (672, 1024)]
    nb = res // 8
    work = [0.0] * nb
    for L in range(0, 84):                       # the work sits in layers 0 .. 84 x 8: plan b spreads it better than plan a would for a shifted camera
        work[L] = 1.0
    for L in range(40, 70):
        work[L] = 3.0
    better, worse = (a, b) if PL.busiest_rank_work(a, work, halo) < PL.busiest_rank_work(b, work, halo) else (b, a)
    pays, gain_s, cost_s = PL.migration_pays(worse, better, work, halo, res, frame_s=0.35e-3, horizon_frames=20)
    assert gain_s > 0 and cost_s > 10e-3 and not pays                       # ~0.7 GB over one link: tens of milliseconds against a millisecond of gain
    assert PL.migration_pays(worse, better, work, halo, res, frame_s=0.35e-3, horizon_frames=200000)[0]
    assert PL.migration_pays([(0, 96), (96, 128)], [(0, 64), (64, 128)], [4.0] * 4 + [1.0] * 12, 0, 128, frame_s=0.2e-3, horizon_frames=50)[0]   # 128^3: 128 KiB per layer
    assert PL.migration_pays(better, worse, work, halo, res, frame_s=0.35e-3, horizon_frames=10**9) == (False, 0.0, 0.0)
    assert PL.migration_pays(a, a, work, halo, res, frame_s=1.0, horizon_frames=100) == (False, 0.0, 0.0)


# ---- dynamic slab boundaries: layers change their owner over the wire ---------------------------------------------------------------------------
def test_migration_plan_covers_exactly_what_a_rank_lacks():
    """plan_migration: every layer of a rank's NEW stored range (own + halo) that it did not store before arrives exactly once, from the rank that
    owned it; nothing else moves; the plan is the same whoever computes it."""
    rng = np.random.default_rng(5)
    for res, world, halo in ((128, 2, 8), (128, 3, 8), (256, 4, 16), (512, 8, 8), (64, 2, 0)):
        nb = res // 8
        for _ in range(40):
            cuts = lambda: [0] + sorted(rng.choice(np.arange(1, nb), world - 1, replace=False).tolist()) + [nb]
            a, b = cuts(), cuts()
            old = [(a[i] * 8, a[i + 1] * 8) for i in range(world)]
            new = [(b[i] * 8, b[i + 1] * 8) for i in range(world)]
            plan = PL.plan_migration(old, new, halo, res)
            assert plan == PL.plan_migration(list(old), list(new), halo, res)
            got = {d: set() for d in range(world)}
            for s, d, z0, z1 in plan:
                assert s != d and z0 % 8 == 0 and z1 % 8 == 0 and z0 < z1
                assert old[s][0] <= z0 and z1 <= old[s][1]                      # the source owned every layer of the piece
                layers = set(range(z0, z1))
                assert not (layers & got[d])
                got[d] |= layers
            for d in range(world):
                o, n = PL.stored_range(old[d], halo, res), PL.stored_range(new[d], halo, res)
                assert got[d] == set(range(n[0], n[1])) - set(range(o[0], o[1]))


def _migration_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import slab_cpu_ops as ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, halo = 192, 8
    ranges = PL.slab_ranges(res, world)
    slab = ops.FakeSlab(res, ranges[rank], halo)
    mig = PL.SlabMigrator(dist, rank, world, res, halo, slab.export, slab.import_, slab.resize, slab.alloc)
    rng = np.random.default_rng(11)                                             # identical on every rank
    ok, moved_total = True, 0
    nb = res // 8
    for step in range(6):
        slab.integrate(step)                                                     # every rank fuses the frame into what it stores
        if step == 3:                                                            # the boundaries the balancer itself would pick for a far-heavy workload
            work = np.where(np.arange(nb) > nb // 2, 5.0, 0.2)
            new = PL.slab_ranges(res, world, work.tolist(), halo=halo)
        else:
            cuts = [0] + sorted(rng.choice(np.arange(1, nb), world - 1, replace=False).tolist()) + [nb]
            new = [(cuts[i] * 8, cuts[i + 1] * 8) for i in range(world)]
        plan, moved = mig.migrate(ranges, new)
        moved_total += moved
        ranges = new
        ok = ok and slab.owned == tuple(new[rank]) and slab.stored == PL.stored_range(new[rank], halo, res)
        ok = ok and np.array_equal(slab.data.view(np.uint32), ops.FakeSlab.truth(slab.stored[0], slab.stored[1], slab.xy, step).view(np.uint32))
    flag = torch.tensor([1 if ok else 0, moved_total], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    open(os.path.join(out_dir, "mig%d.txt" % rank), "w").write("ok %d" % flag[1].item() if flag[0].item() else "mismatch")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_layers_migrate_between_ranks_gloo(tmp_path, world):
    """SlabMigrator over a gloo group: after every change of the boundaries (random ones, and the balancer's own) each rank holds, bit for bit, what
    it would hold had it integrated its new stored range from the start -- own layers and halo, whichever rank they came from."""
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_migration_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        txt = open(os.path.join(str(tmp_path), "mig%d.txt" % r)).read()
        assert txt.startswith("ok"), (r, txt)
