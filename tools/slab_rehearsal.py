#!/usr/bin/env python3
"""Rehearsal of the N-rank z-slab pipeline on a box with ONE GPU: `python tools/slab_rehearsal.py [--ranks 2]`.

The parent touches no GPU; it starts N rank processes (torch.distributed.run).  The ranks share GPU 0 and form a gloo group
(RCCL refuses two ranks on one device), run SlabPipeline -- real kernels, real collectives, the launcher path bench.py uses --
for a few frames, and every rank compares against SingleGpuPipeline run in the same process: tracked poses and merged model
maps bit for bit, its owned volume layers bit for bit.  Prints `REHEARSAL OK ranks=N` from rank 0 on success.

--rebalance-every K: the slab boundaries follow the work (SlabPipeline.rebalance: work per brick layer counted by the fusion pass, pooled, slab_ranges
re-run, layers that change owner sent rank to rank) -- with --yaw D / --dolly M the camera turns by D degrees and moves M metres into the scene over
the sequence, so the work shifts along z and migrations happen; the comparison with the single-GPU pipeline then covers frames before and after them."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    import numpy as np
    import torch
    import torch.distributed as dist
    from hybkinectfu_amd import lib as K
    from hybkinectfu_amd import pipeline as PL
    from hybkinectfu_amd import scene as S
    P = S.STOCK
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    res, size, cam = a.res, a.size, S.vga_camera()
    wl = dict(trunc_max=P["depth_trunc_max"], integ_dist=P["integrate_depth_trunc"])
    n = a.frames
    def pose_of(k):                                       # the stock circle, plus a slow turn and a dolly into the scene (the work moves along z)
        p = S.trajectory_pose(k, size)
        f = k / max(n - 1, 1)
        yaw = np.radians(a.yaw) * f
        turn = np.array([[np.cos(yaw), 0, np.sin(yaw), 0], [0, 1, 0, 0], [-np.sin(yaw), 0, np.cos(yaw), a.dolly * f], [0, 0, 0, 1.0]])
        return p @ turn
    frames = np.stack([S.render_depth_mm(pose_of(k), cam, size) for k in range(n)])
    dev = torch.from_numpy(frames.astype(np.int16)).cuda()
    fb = cam[0] * cam[1] * 2
    outs = []
    trace = {}
    for cls in (PL.SingleGpuPipeline, PL.SlabPipeline):
        pipe = (cls(K.camera(*cam), res, size, wl) if cls is PL.SingleGpuPipeline else
                cls(K.camera(*cam), res, size, wl, rank=rank, world=world, rebalance_every=a.rebalance_every, rebalance_sample=2))
        poses = []
        for k in range(n):
            nxt = dev.data_ptr() + (k + 1) * fb if (k % 2 == 0 and k + 1 < n) else None
            pipe.process_frame_device(dev.data_ptr() + k * fb, k, nxt)
            ok, pose, _, _ = pipe.track_result()
            assert ok, "rank %d lost frame %d" % (rank, k)
            poses.append(pose.copy())
            if a.trace:                                   # per-frame fingerprints of the merged model maps and of the stored volume layers' update count
                import hashlib
                hm = hashlib.sha1(pipe.ctx.download_map(K.MAP_MODEL_VERTICES).tobytes() + pipe.ctx.download_map(K.MAP_MODEL_NORMALS).tobytes()).hexdigest()[:12]
                trace.setdefault(cls.__name__, []).append((hm, int(pipe.stats()["updated_last"])))
        pipe.sync()
        maps = [pipe.ctx.download_map(m) for m in (K.MAP_MODEL_VERTICES, K.MAP_MODEL_NORMALS)]
        z0, z1 = pipe.ctx.owned
        vol = pipe.ctx.download_volume(z0, z1)
        outs.append((poses, maps, vol, (z0, z1), pipe.stats()["updated_total"]))
        if cls is PL.SlabPipeline:
            for (f, old, new, moved) in pipe.migrations:
                print("rank %d: frame %d boundaries %s -> %s, %d voxel layers sent / received here" % (rank, f, old, new, moved), flush=True)
            n_mig = len(pipe.migrations)
        pipe.close()
    (p1, m1, v1, _, _), (p2, m2, v2, (z0, z1), upd) = outs
    pose_ok = [bool(np.array_equal(a_, b_)) for a_, b_ in zip(p1, p2)]
    maps_ok = all(np.array_equal(a_.view(np.uint32), b_.view(np.uint32)) for a_, b_ in zip(m1, m2))
    vol_ok = np.array_equal(v1[0][z0:z1].view(np.uint32), v2[0].view(np.uint32)) and np.array_equal(v1[1][z0:z1], v2[1])
    ok = all(pose_ok) and maps_ok and vol_ok
    if not ok:
        bad_layers = [z0 + int(i) for i in np.nonzero((v1[0][z0:z1].view(np.uint32) != v2[0].view(np.uint32)).reshape(z1 - z0, -1).any(axis=1) |
                                                      (v1[1][z0:z1] != v2[1]).reshape(z1 - z0, -1).any(axis=1))[0]]
        print("rank %d: first pose mismatch at frame %s, maps %s, volume %s (layers %s)" % (
            rank, pose_ok.index(False) if False in pose_ok else None, "ok" if maps_ok else "MISMATCH", "ok" if vol_ok else "MISMATCH",
            (bad_layers[:4] + ["..."] + bad_layers[-2:]) if len(bad_layers) > 6 else bad_layers), flush=True)
    if a.trace:
        ta, tb = trace["SingleGpuPipeline"], trace["SlabPipeline"]
        first_map = next((k for k in range(n) if ta[k][0] != tb[k][0]), None)
        if rank == 0:
            print("trace: first frame whose merged model maps differ from the single-GPU pipeline's: %s; first frame whose pose differs: %s" % (
                first_map, pose_ok.index(False) if False in pose_ok else None), flush=True)
    valid = int((m2[0][..., 3] != 0).sum())
    if a.rebalance_every and a.expect_migration and n_mig == 0:
        ok = False
        print("rank %d: no migration happened" % rank, flush=True)
    flag = torch.tensor([1 if ok else 0, valid], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    print("rank %d: slab z[%d,%d) poses/maps/volume %s, %d valid model pixels, %d voxel updates" % (rank, z0, z1, "bit-exact" if ok else "MISMATCH", valid, upd), flush=True)
    if rank == 0:
        print(("REHEARSAL OK" if flag[0].item() == 1 and flag[1].item() > 10000 else "REHEARSAL FAILED") + " ranks=%d res=%d frames=%d" % (world, res, n), flush=True)
    dist.destroy_process_group()
    return 0 if flag[0].item() == 1 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--res", type=int, default=384)
    ap.add_argument("--size", type=float, default=3.0)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--rebalance-every", type=int, default=0)
    ap.add_argument("--yaw", type=float, default=0.0, help="degrees the camera turns over the sequence")
    ap.add_argument("--dolly", type=float, default=0.0, help="metres the camera moves into the scene over the sequence")
    ap.add_argument("--trace", action="store_true", help="per-frame fingerprints of the merged model maps: where does a mismatch start?")
    ap.add_argument("--expect-migration", action="store_true", help="fail when the run ends without a single migration")
    a = ap.parse_args()
    if "WORLD_SIZE" in os.environ:
        sys.exit(worker(a))
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.ranks),
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))


if __name__ == "__main__":
    main()
