#!/usr/bin/env python3
"""Per-rank GPU time of the z-slab pipeline at world size N, measured on ONE GPU: rank r's slab context runs alone and its
collectives run on a one-rank group, so the number is the compute part of a frame at N GPUs (collective time not included).
usage: tools/slab_rank_time.py [world] [rank ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from hybkinectfu_amd import lib as K, scene as S
from hybkinectfu_amd.pipeline import SlabPipeline
import bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(a) for a in sys.argv[2:]] or [0, world // 2, world - 1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
wl = bench.workload(world, "c4")
cam = wl["cam"]
frames, _ = S.make_stream(60, cam, wl["size"])
dev = torch.from_numpy(frames.astype(np.int16)).cuda()
fb = cam[0] * cam[1] * 2
for r in ranks:
    pipe = SlabPipeline(K.camera(*cam), wl["res"], wl["size"], wl, rank=r, world=world, device=0)
    for k in range(10):
        pipe.process_frame_device(dev.data_ptr() + k * fb, k)
    pipe.sync()
    pipe.stage_timers(0x1E)
    t0 = time.perf_counter()
    for k in range(10, 60):
        pipe.process_frame_device(dev.data_ptr() + k * fb, k)
    pipe.sync()
    dt = (time.perf_counter() - t0) / 50
    ms, cnt = pipe.read_stage_ms()
    print("world %d rank %d slab %s: %.3f ms/frame; stages pre %.3f track %.3f integrate %.3f raycast %.3f" % (
        world, r, pipe.slab, dt * 1e3, ms[1] / cnt[1], ms[2] / cnt[2], ms[3] / cnt[3], ms[4] / cnt[4]))
    pipe.close()
dist.destroy_process_group()
