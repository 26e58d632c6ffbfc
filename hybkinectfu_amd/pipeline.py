"""Per-frame pipelines over the C ABI, mirroring HybKinectfu::processNewFrame (src/HybKinectfu.cpp:98-160).

SingleGpuPipeline: one context, the whole volume.  Everything a frame needs is enqueued on the context's stream with
no host synchronisation: the pose and the tracking verdict stay on the device (kf_icp_track), integrate is predicated on
the verdict inside the kernel (src/HybKinectfu.cpp:123), and raycast reads the device-resident pose.
"""
import os

from . import lib as K
from . import scene as S

P = S.STOCK


class SingleGpuPipeline:
    def __init__(self, kcam, res, size, wl=None, device=0, max_triangles=0):
        wl = wl or {}
        self.cam = kcam
        self.trunc_max = wl.get("trunc_max", P["depth_trunc_max"])
        self.integ_dist = wl.get("integ_dist", P["integrate_depth_trunc"])
        self.tracker = wl.get("tracker", "icp")                # "icp": CameraPoseFinderICP, "sdf": CameraPoseFinderSDF
        self.color = bool(wl.get("color", False))              # the reference's stock switches use_color=1, color_angle_weight=1 (src/config.ini:2,7)
        self.ctx = K.Context(kcam, res, size, wl.get("max_weight", P["volume_max_weight"]), levels=3, max_triangles=max_triangles, device=device, has_color=self.color)
        self.ctx.set_pose(S.pose0(size))                       # HybKinectfu::init  src/HybKinectfu.cpp:51-57
        self.inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]     # AppParamsProducer.cpp:113-117
        self._host = None                                      # process_frame_host: what is staged ahead

    def process_frame_device(self, dev_mm_ptr, frame_id, next_mm_ptr=None):
        """next_mm_ptr: where the NEXT frame already lies in HBM (streaming input): its depth conversion + gate + bilateral filter ride in this
        frame's tracking launch and its vertices / normals in this frame's raycast launch (or, when the tracker takes another launch form, the
        filter rides in the raycast launch and the vertices / normals follow it)."""
        c = self.ctx
        c.set_depth_mm_device(dev_mm_ptr)                                                                    # copyFrameToGPU
        c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])  # :106-110
        if next_mm_ptr is not None:           # before the tracker: the next frame's filter rides on the CUs the ICP loop leaves idle (else: in the raycast launch)
            c.prefetch_frame(next_mm_ptr, P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        if self.tracker == "sdf":
            c.sdf_track(frame_id, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])                  # :116 with CameraPoseFinderSDF
        else:
            c.icp_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])  # :116
        c.integrate(None, P["integrate_sdf_trunc"], self.integ_dist, has_color=self.color, angle_weight=self.color)      # :125-140
        c.raycast(None, self.inc, P["depth_trunc_min"], self.trunc_max, has_color=self.color)                           # :149-154

    def process_frame_host(self, frame_of, frame_id):
        """A stream whose frames start in HOST memory (HybKinectfu::copyFrameToGPU, src/HybKinectfu.cpp:63-96): `frame_of(k)` -> u16 millimetres of
        frame k, or None past the end of the stream.  Frame k + 2 crosses PCIe (kf_upload_depth_mm_next, a free upload slot) while frame k is
        processed and frame k + 1's front end rides in frame k's launches: the copy is off the critical path.  Frame ids that do not follow the
        previous call's by one restart the staging (that frame is then uploaded and waited for)."""
        c = self.ctx
        st = self._host
        if st is None or st["expect"] != frame_id or st["staged"] == 0:
            c.upload_depth_mm(frame_of(frame_id))              # (drops whatever was staged)
            st = self._host = dict(expect=frame_id, staged=0, next_dev=None)
            nxt = frame_of(frame_id + 1)
            if nxt is not None:
                st["next_dev"], st["staged"] = c.upload_depth_mm_next(nxt), 1
        else:
            c.take_next_depth()
            st["staged"] -= 1
        c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        dev_after = None
        if st["staged"] == 1:
            after = frame_of(frame_id + 2)
            if after is not None:
                dev_after, st["staged"] = c.upload_depth_mm_next(after), 2
            c.prefetch_frame(st["next_dev"], P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        st["next_dev"], st["expect"] = dev_after, frame_id + 1
        if self.tracker == "sdf":
            c.sdf_track(frame_id, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"])
        else:
            c.icp_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        c.integrate(None, P["integrate_sdf_trunc"], self.integ_dist, has_color=self.color, angle_weight=self.color)
        c.raycast(None, self.inc, P["depth_trunc_min"], self.trunc_max, has_color=self.color)

    def sync(self):
        self.ctx.sync()

    def stats(self, observed=False):
        """the fusion passes' counters (updated voxels, frames fused / lost, queued bricks); observed=True adds weight_gt0 (kf_get_volume_stats)"""
        return self.ctx.stats(observed)

    def stage_timers(self, mask):
        self.ctx.stage_timers(mask)

    def read_stage_ms(self):
        return self.ctx.read_stage_ms()

    def track_result(self):
        return self.ctx.track_result()

    def close(self):
        self.ctx.close()


# ---------------------------------------------------------------------------------------------------------------------------
# z-slab partitioning over the GPUs of one node: one process per GPU, torch.distributed over RCCL.
# ---------------------------------------------------------------------------------------------------------------------------


def slab_ranges(res, world, layer_work=None, halo=0):
    """Owned z range [z0, z1) of every rank: contiguous, brick (8-layer) aligned, every rank at least one brick layer.

    Without `layer_work`: equal thickness (sizes differing by at most one brick).  With it -- the work (voxels fused per frame) of every
    8-voxel brick layer, e.g. from probe_layer_work -- the boundaries that minimise the BUSIEST rank's work, halo layers (re-integrated by
    both neighbours) included: equal z-slabs are unequally busy (the nearest sees a narrow frustum, the farthest may lie behind every
    surface: 15 / 45 / 38 / 2 % on four ranks at 1024^3 @ 6 m), and a frame lasts as long as the busiest rank needs.  Exact: a dynamic
    programme over the (rank, boundary) pairs; deterministic, so every rank derives the same ranges from the same probe."""
    nb = res // 8
    if world > nb:
        raise ValueError("more ranks (%d) than brick layers (%d)" % (world, nb))
    if layer_work is None:
        base, extra = divmod(nb, world)
        out, b = [], 0
        for r in range(world):
            n = base + (1 if r < extra else 0)
            out.append((b * 8, (b + n) * 8))
            b += n
        return out
    w = [float(x) for x in layer_work]
    if len(w) != nb:
        raise ValueError("layer_work needs one entry per brick layer (%d), got %d" % (nb, len(w)))
    hb = (int(halo) + 7) // 8
    pre = [0.0]
    for x in w:
        pre.append(pre[-1] + x)
    eps = 1e-9 * (pre[-1] + 1.0)          # a tiny cost per layer: ties resolve towards even thickness, empty regions are still shared out

    def cost(i, j):                       # work of a rank owning brick layers [i, j): own + halo layers
        lo, hi = max(0, i - hb), min(nb, j + hb)
        return pre[hi] - pre[lo] + eps * (j - i) * (j - i)
    INF = float("inf")
    best = [[INF] * (nb + 1) for _ in range(world + 1)]
    cut = [[0] * (nb + 1) for _ in range(world + 1)]
    best[0][0] = 0.0
    for r in range(1, world + 1):
        for j in range(r, nb - (world - r) + 1):
            for i in range(r - 1, j):
                if best[r - 1][i] == INF:
                    continue
                c = max(best[r - 1][i], cost(i, j))
                if c < best[r][j]:
                    best[r][j], cut[r][j] = c, i
    bounds, j = [nb], nb
    for r in range(world, 0, -1):
        j = cut[r][j]
        bounds.append(j)
    bounds.reverse()
    return [(bounds[r] * 8, bounds[r + 1] * 8) for r in range(world)]


def probe_layer_work(kcam, res, size, wl, dev_mm_ptr, device=0, probe_res=256):
    """Voxels one frame fuses per 8-voxel brick layer of the res^3 volume, estimated from ONE integrate of that frame into a low-resolution
    volume of the same extent (probe_res^3: a few milliseconds and a 67 MB read-back at 256^3) -- the input of slab_ranges' balancing.
    Every rank runs the same probe on the same frame and gets the same numbers."""
    import numpy as np
    wl = wl or {}
    pres = min(int(probe_res), int(res))
    ctx = K.Context(kcam, pres, size, P["volume_max_weight"], levels=3, device=device)
    ctx.set_pose(S.pose0(size))
    ctx.set_depth_mm_device(dev_mm_ptr)
    ctx.preprocess(P["depth_trunc_min"], wl.get("trunc_max", P["depth_trunc_max"]), P["filter_sigma_pixel"], P["filter_sigma_depth"])
    ctx.integrate(S.pose0(size), P["integrate_sdf_trunc"], wl.get("integ_dist", P["integrate_depth_trunc"]))
    _, w = ctx.download_volume()
    ctx.close()
    per_layer = (w > 0).reshape(pres, -1).sum(axis=1).astype(np.float64)           # voxels fused per probe z-layer
    nb = res // 8
    scale = (res / float(pres)) ** 3                                               # voxel count at full resolution
    edges = np.linspace(0, pres, nb + 1)
    cum = np.concatenate([[0.0], np.cumsum(per_layer)])
    work = np.diff(np.interp(edges, np.arange(pres + 1), cum)) * scale
    return work.tolist()


def merge_candidates(t, v, n, all_reduce_min, all_reduce_sum_i32):
    """First crossing along each ray wins (the reference's sequential march, raycastingVolume.cu:83-116).

    t [H,W] float32 (+inf = no crossing in this slab), v / n [H,W,4] float32 candidates of this rank.  A ray sample belongs to
    exactly one slab, so the minimum t has a unique owner; everyone else contributes zero bits and an INTEGER sum returns the
    winner's vertex/normal bit for bit (a float sum would turn -0.0 into +0.0).  Works on any backend (RCCL on GPU, gloo on CPU).
    """
    import torch
    tmin = t.clone()
    all_reduce_min(tmin)
    win = (t == tmin) & torch.isfinite(t)
    vi = v.view(torch.int32) * win.unsqueeze(-1).to(torch.int32)
    ni = n.view(torch.int32) * win.unsqueeze(-1).to(torch.int32)
    all_reduce_sum_i32(vi)
    all_reduce_sum_i32(ni)
    return vi.view(torch.float32), ni.view(torch.float32)


class SlabExchange:
    """The collective sequence that merges one frame's per-slab raycast results (first crossing along each ray wins):

        MIN all-reduce(ta)                     ta[px] = (crossing parameter << 32 | vertex parameter alpha) as int64: positive floats order like their bits,
          (asynchronous around overlap()       so the first crossing along the ray wins and brings its alpha along
           when a caller passes one)
        normals(ta, cand)                      every rank rebuilds the winners' vertices from the rays; the OWNER of a vertex's layer writes the normal (3 words)
        integer SUM all-reduce(cand bits)      exactly one rank contributes non-zero bits per pixel; integer sums keep -0.0
        unpack(ta, cand)                       -> model maps (+ pyramids) of every rank

    The gradient is the vertex owner's job, not the crossing owner's: the reference's vertex is an extrapolation (raycastingVolume.cu:89-90) that can land
    far from the crossing, outside that slab's halo.  `normals` / `unpack` are device launches (kf_slab_ray_normals / kf_set_model_maps_rays) in
    SlabPipeline; the CPU-only tests and `bench.py --collective-selftest` pass restatements and a gloo group, so the SAME sequence of collectives runs
    at world_size 2 / 3 without a GPU.  8 + 12 bytes per pixel on the wire.
    """

    def __init__(self, rows, cols, device, dist, normals, unpack):
        import torch
        self.dist, self.normals, self.unpack = dist, normals, unpack
        self.ta = torch.empty((rows, cols), dtype=torch.int64, device=device)              # this rank's crossings, then (in place) the winners
        self.cand = torch.empty((rows, cols, 3), dtype=torch.float32, device=device)       # the normal where this rank owns the winner's vertex (all-zero bits elsewhere: a found normal is a unit vector)
        self.cand_bits = self.cand.view(torch.int32)

    def merge(self, overlap=None):
        dist = self.dist
        if overlap is None:
            # nothing to overlap: a synchronous collective is enqueued on the CURRENT stream (the pipeline's own: torch >= 2.8 ProcessGroupNCCL, asyncOp = false) --
            # an asynchronous one runs on RCCL's internal stream, with an event hand-off into it and another one back (~15 us of stream time per collective)
            dist.all_reduce(self.ta, op=dist.ReduceOp.MIN)
        else:
            pending = dist.all_reduce(self.ta, op=dist.ReduceOp.MIN, async_op=True)
            overlap()             # runs while the collective is in flight (RCCL's own stream until wait() joins it)
            pending.wait()
        self.normals(self.ta, self.cand)
        dist.all_reduce(self.cand_bits, op=dist.ReduceOp.SUM)
        self.unpack(self.ta, self.cand)


def slab_halo_layers(res, size, ray_increment):
    """Voxel layers a slab stores AND integrates beyond its owned range, per side: ceil(inc/voxel) + 2,
    rounded up to whole 8-layer bricks.  With x = inc/voxel: the previous ray sample lies at most x layers outside the owned
    range (raycast.hip reads it from the halo), its trilinear taps reach floor(-x - 0.5) and the gradient taps around a vertex
    next to it floor(-x - 1.5) >= -(ceil(x) + 2).  kf_raycast_volume_slab refuses a context whose halo is thinner."""
    import math
    need = int(math.ceil(ray_increment * res / size)) + 2
    return ((need + 7) // 8) * 8


def stored_range(owned, halo, res):
    """brick-aligned layers a rank with owned range `owned` stores: own + halo per side, clipped to the volume (kf_create / kf_resize_slab)"""
    hb = (int(halo) + 7) // 8
    return (max(0, owned[0] - 8 * hb), min(res, owned[1] + 8 * hb))


def busiest_rank_work(ranges, layer_work, halo):
    """work (own + halo brick layers) of the busiest rank under `ranges` -- what slab_ranges minimises"""
    hb = (int(halo) + 7) // 8
    nb = len(layer_work)
    return max(sum(layer_work[max(0, a // 8 - hb):min(nb, b // 8 + hb)]) for a, b in ranges)


def plan_migration(old_ranges, new_ranges, halo, res):
    """Which voxel layers move between which ranks when the owned ranges change from old_ranges to new_ranges: a list of (src, dst, z0, z1),
    brick aligned, in ONE canonical order every rank derives for itself (no negotiation).  A rank needs every layer of its NEW stored range
    (own + halo) that it did not store before; the source of a layer is the rank that OWNED it before (its halo copies on the neighbours are
    bit-identical -- both re-integrate them -- but the owner is the one deterministic choice).  Consecutive layers with the same source travel
    as one piece."""
    world = len(old_ranges)
    owner_of = {}
    for r, (a, b) in enumerate(old_ranges):
        for L in range(a // 8, b // 8):
            owner_of[L] = r
    plan = []
    for d in range(world):
        o0, o1 = stored_range(old_ranges[d], halo, res)
        n0, n1 = stored_range(new_ranges[d], halo, res)
        run = None                                              # (src, first layer, one past the last)
        for L in range(n0 // 8, n1 // 8):
            src = None if o0 // 8 <= L < o1 // 8 else owner_of[L]
            if run is not None and (src != run[0] or src is None):
                plan.append((run[0], d, run[1] * 8, run[2] * 8)); run = None
            if src is not None:
                run = (src, L, L + 1) if run is None else (run[0], run[1], L + 1)
        if run is not None:
            plan.append((run[0], d, run[1] * 8, run[2] * 8))
    return sorted(plan, key=lambda t: (t[2], t[1], t[0]))


def migration_pays(old_ranges, new_ranges, work, halo, res, frame_s, horizon_frames, shard_share=0.5, link_gbps=40.0):
    """Does moving the slab boundaries pay for itself before the next decision?  Gain: the busiest rank's work drops from W_old to W_new, and
    `shard_share` of a frame (fusion + slab raycast; the replicated tracker and the merge do not shrink) lasts as long as the busiest rank needs:
    (1 - W_new / W_old) * shard_share * frame_s per frame, for `horizon_frames` frames.  Cost: the rank that receives the most voxel layers waits
    for them -- res^2 voxels x 8 bytes per layer over one xGMI link (`link_gbps`; point-to-point, ~48 GB/s each way on MI355X, less in practice).
    Returns (pays, gain_s, cost_s).  Pure: every rank computes the same verdict from the same (pooled) inputs."""
    w_old, w_new = busiest_rank_work(old_ranges, work, halo), busiest_rank_work(new_ranges, work, halo)
    if w_old <= 0.0 or w_new >= w_old:
        return False, 0.0, 0.0
    gain_s = (1.0 - w_new / w_old) * shard_share * frame_s * horizon_frames
    received = [0] * len(old_ranges)
    for (_, dst, z0, z1) in plan_migration(old_ranges, new_ranges, halo, res):
        received[dst] += z1 - z0
    cost_s = max(received) * float(res) * float(res) * 8.0 / (link_gbps * 1e9)
    return gain_s > cost_s, gain_s, cost_s


class SlabMigrator:
    """Moves voxel layers between ranks when the slab boundaries change (the xGMI exchange of boundary slabs the design needs: a long sequence
    that turns the camera shifts the work along z, and a frame lasts as long as the busiest rank needs).

        plan = plan_migration(old, new)            the same list on every rank
        senders export their pieces                BEFORE their own resize may drop them
        resize(new owned range)                    layers stored before and after keep their voxels, new ones start unobserved
        for every piece, in plan order:            dist.send on the source, dist.recv + import on the destination

    export(z0, z1) -> tensor, import_(z0, z1, tensor), resize(z0, z1) and alloc(z0, z1) -> empty tensor are the rank's volume operations
    (SlabPipeline: kf_download_volume_device / kf_upload_volume_device / kf_resize_slab on torch CUDA tensors; the CPU-only tests: a numpy
    stand-in), wire / unwire convert a piece to and from what the process group can carry (gloo: CPU tensors).  Blocking sends and receives
    in one global order cannot deadlock: the earliest unfinished piece always has both its ends free."""

    def __init__(self, dist, rank, world, res, halo, export, import_, resize, alloc, wire=None, unwire=None):
        self.dist, self.rank, self.world, self.res, self.halo = dist, rank, world, res, halo
        self.export, self.import_, self.resize, self.alloc = export, import_, resize, alloc
        self.wire = wire or (lambda t: t)
        self.unwire = unwire or (lambda t: t)

    def migrate(self, old_ranges, new_ranges):
        plan = plan_migration(old_ranges, new_ranges, self.halo, self.res)
        outgoing = {i: self.wire(self.export(z0, z1)) for i, (s, d, z0, z1) in enumerate(plan) if s == self.rank}
        self.resize(*new_ranges[self.rank])
        moved = 0
        for i, (s, d, z0, z1) in enumerate(plan):
            if s == self.rank:
                self.dist.send(outgoing.pop(i), dst=d)
                moved += z1 - z0
            elif d == self.rank:
                buf = self.wire(self.alloc(z0, z1))
                self.dist.recv(buf, src=s)
                self.import_(z0, z1, self.unwire(buf))
                moved += z1 - z0
        return plan, moved


class SlabPipeline:
    """One rank of the z-slab partitioned pipeline.

    Replicated per rank (cheap, and bitwise identical everywhere): preprocess, pyramids, the whole ICP loop.
    Partitioned: TSDF integrate (own slab + halo, no communication: the halo layers are RE-INTEGRATED by both neighbours
    instead of exchanged -- integration is a pure function of voxel, depth and pose) and raycast (own samples only), followed
    by SlabExchange's two all-reduces.
    """

    def __init__(self, kcam, res, size, wl=None, rank=0, world=1, device=0, max_triangles=0, icp_mode="replicated", tracker="icp", ranges=None,
                 rebalance_every=0, rebalance_gain=0.10, rebalance_sample=4, link_gbps=40.0, shard_share=0.5):
        import torch
        import torch.distributed as dist
        self.icp_mode = icp_mode            # "replicated" (default, faster at VGA) or "allreduce" (pixels split over the ranks)
        self.tracker = tracker              # "icp" (CameraPoseFinderICP) or "sdf" (CameraPoseFinderSDF: pixels go to the slab that owns their world point, 27-float all-reduce per iteration)
        wl = wl or {}
        self.dist, self.torch = dist, torch
        self.rank, self.world = rank, world
        self.trunc_max = wl.get("trunc_max", P["depth_trunc_max"])
        self.integ_dist = wl.get("integ_dist", P["integrate_depth_trunc"])
        self.inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
        self.halo = slab_halo_layers(res, size, self.inc)
        self.ranges = list(ranges) if ranges is not None else slab_ranges(res, world)        # ranges: e.g. slab_ranges(res, world, probe_layer_work(...), halo)
        self.slab = self.ranges[rank]
        self.ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=max_triangles, device=device,
                             slab=self.slab, halo=self.halo)
        self.ctx.set_pose(S.pose0(size))
        dev = torch.device("cuda", device)
        # ONE stream orders the library's kernels, torch's elementwise ops and the RCCL collectives (which synchronise with the
        # CURRENT torch stream).  It has to be a stream of its own: torch's default stream has the null handle, which
        # kf_set_stream reads as "back to the private stream" -- the kernels would then race with the collectives.
        self.stream = torch.cuda.Stream(device=dev)
        assert self.stream.cuda_stream != 0
        self.ctx.set_stream(self.stream.cuda_stream)
        c = self.ctx
        # speculative normals (include/hybkf.h: kf_raycast_volume_slab_cross_spec): the marching launch evaluates the gradient of this rank's own crossings where it owns
        # the vertex, keeps its own words in ta_own (ex.ta is all-reduced in place) and the gradients in spec; the normals launch copies them where the rank's crossing
        # won and evaluates only what is left.  KF_SLAB_SPECULATE=0: the two-launch form without it (A/B).
        self.speculate = os.environ.get("KF_SLAB_SPECULATE", "1") != "0"
        self.ta_own = torch.empty((kcam.rows, kcam.cols), dtype=torch.int64, device=dev)
        self.spec = torch.empty((kcam.rows, kcam.cols, 3), dtype=torch.float32, device=dev)

        def normals(ta, cand):
            if self.speculate:
                c.slab_ray_normals_spec(None, self.inc, P["depth_trunc_min"], self.trunc_max, ta.data_ptr(), self.ta_own.data_ptr(), self.spec.data_ptr(), cand.data_ptr())
            else:
                c.slab_ray_normals(None, self.inc, P["depth_trunc_min"], self.trunc_max, ta.data_ptr(), cand.data_ptr())
        self.ex = SlabExchange(kcam.rows, kcam.cols, dev, dist, normals=normals,
                               unpack=lambda ta, cand: c.set_model_maps_rays(None, ta.data_ptr(), cand.data_ptr()))
        self.sums = torch.zeros(32, dtype=torch.float32, device=dev)
        self._frame_ns, self._pooled_frame_s = 0, 0.0
        self._merge_events = None           # time_merge(True): (start, end) torch event pairs around every frame's merge
        # dynamic re-balancing of the slab boundaries: every `rebalance_every` frames the ranks pool the work per brick layer that the fusion pass
        # counted over the last `rebalance_sample` frames, re-run slab_ranges (deterministic: every rank computes the same boundaries) and, when the
        # busiest rank's work drops by more than `rebalance_gain`, move the layers that change owner (SlabMigrator).  0: boundaries stay as given.
        self.res, self.size, self.dev = int(res), float(size), dev
        self.rebalance_every, self.rebalance_gain, self.rebalance_sample = int(rebalance_every), float(rebalance_gain), max(1, int(rebalance_sample))
        self.migrations = []                # [(frame, old ranges, new ranges, voxel layers this rank sent or received)]
        self.link_gbps, self.shard_share = float(link_gbps), float(shard_share)      # the cost side of a migration (migration_pays)
        self._last_decision = None          # (wall clock, frame) of the previous re-balance decision: the frame time comes from there
        self.declined = []                  # [(frame, gain_s, cost_s)] plans that would not have paid for themselves
        cpu_wire = world > 1 and dist.is_initialized() and dist.get_backend() == "gloo"       # rehearsals: gloo carries CPU tensors only
        self.migrator = SlabMigrator(dist, rank, world, self.res, self.halo, export=self._export_layers, import_=self._import_layers,
                                     resize=lambda z0, z1: c.resize_slab(z0, z1, self.halo),
                                     alloc=lambda z0, z1: torch.empty((2, z1 - z0, self.res, self.res), dtype=torch.float32, device=dev),
                                     wire=(lambda t: t.cpu()) if cpu_wire else None, unwire=(lambda t: t.to(dev)) if cpu_wire else None)

    def process_frame_device(self, dev_mm_ptr, frame_id, next_mm_ptr=None):
        with self.torch.cuda.stream(self.stream):
            self._process_frame_device(dev_mm_ptr, frame_id, next_mm_ptr)

    def _preprocess(self, dev_mm_ptr):
        c = self.ctx
        c.set_depth_mm_device(dev_mm_ptr)
        c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])

    # ---- slab boundaries that follow the work ----
    def _export_layers(self, z0, z1):
        t = self.torch.empty((2, z1 - z0, self.res, self.res), dtype=self.torch.float32, device=self.dev)
        self.ctx.download_volume_device(z0, z1, t[0].data_ptr(), t[1].data_ptr())
        self.stream.synchronize()
        return t

    def _import_layers(self, z0, z1, t):
        t = t.contiguous()
        self.ctx.upload_volume_device(z0, z1, t[0].data_ptr(), t[1].data_ptr())
        self.stream.synchronize()             # (t may be freed by the caller)

    def pooled_layer_work(self):
        """work per brick layer of the WHOLE volume: every layer's count comes from the rank that owns it (a collective)"""
        import numpy as np
        torch, dist = self.torch, self.dist
        mine = torch.from_numpy(np.concatenate([self.ctx.read_layer_work(reset=True).astype(np.int64), [int(self._frame_ns)]]))    # (+ this rank's frame time: pooled too)
        if self.world == 1:
            self._pooled_frame_s = float(mine[-1]) * 1e-9
            return mine.numpy()[:-1].astype(np.float64)
        on_dev = dist.get_backend() != "gloo"
        mine = mine.to(self.dev) if on_dev else mine
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        work = np.zeros(self.res // 8, np.float64)
        every = [e.cpu().numpy() for e in every]
        for r, (a, b) in enumerate(self.ranges):
            work[a // 8:b // 8] = every[r][a // 8:b // 8]
        self._pooled_frame_s = max(float(e[-1]) for e in every) * 1e-9           # the slowest rank's: the same number on every rank
        return work

    def rebalance(self, frame_id=-1, force_ranges=None):
        """A collective: pool the sampled work per brick layer, compute the boundaries that minimise the busiest rank and migrate when that
        pays (or to `force_ranges`, tests).  Returns the new ranges when layers moved, else None."""
        import time
        with self.torch.cuda.stream(self.stream):
            # the frame time since the previous decision (wall clock between two synchronised points), pooled with the work: the slowest rank's counts
            self.sync()
            now = time.perf_counter()
            self._frame_ns = 0
            if self._last_decision is not None and frame_id > self._last_decision[1]:
                self._frame_ns = int((now - self._last_decision[0]) / (frame_id - self._last_decision[1]) * 1e9)
            work = self.pooled_layer_work()
            self._last_decision = (time.perf_counter(), frame_id)
            old = list(self.ranges)
            if force_ranges is not None:
                new = [tuple(r) for r in force_ranges]
            else:
                if work.sum() <= 0:
                    return None
                new = slab_ranges(self.res, self.world, work.tolist(), halo=self.halo)
                if busiest_rank_work(new, work.tolist(), self.halo) > (1.0 - self.rebalance_gain) * busiest_rank_work(old, work.tolist(), self.halo):
                    return None
                if self._pooled_frame_s > 0.0 and self.rebalance_every > 0:          # (no frame time yet -- the first decision -- : the gain threshold alone decides)
                    pays, gain_s, cost_s = migration_pays(old, new, work.tolist(), self.halo, self.res, self._pooled_frame_s, self.rebalance_every,
                                                          self.shard_share, self.link_gbps)
                    if not pays:
                        self.declined.append((frame_id, gain_s, cost_s))
                        return None
            if new == old:
                return None
            self.sync()
            plan, moved = self.migrator.migrate(old, new)
            self.ranges, self.slab = new, new[self.rank]
            self.migrations.append((frame_id, old, new, moved))
            return new

    def _process_frame_device(self, dev_mm_ptr, frame_id, next_mm_ptr):
        c, dist, ex = self.ctx, self.dist, self.ex
        if self.rebalance_every and frame_id > 0:
            if frame_id % self.rebalance_every == 0:
                self.rebalance(frame_id)
            elif (frame_id + self.rebalance_sample) % self.rebalance_every == 0:
                c.count_layer_work(self.rebalance_sample)             # the frames right before the next check are the sample
        self._preprocess(dev_mm_ptr)                                      # (adopts the set the previous frame's launches prepared, if they did)
        if next_mm_ptr is not None:
            # the next frame's front end rides in this frame's launches (kf_prefetch_frame): gate + bilateral in the tracking launch when the persistent
            # loop runs, tile tables + vertices / normals + their pyramids in the raycast launch -- the replicated part of a rank's frame shrinks
            c.prefetch_frame(next_mm_ptr, P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        if self.tracker == "sdf":
            c.sdf_partition_track(frame_id, P["sdf_max_iter_nums"], P["camera_shake_dist"], P["camera_shake_angle"], self.sums.data_ptr(),
                                  lambda: dist.all_reduce(self.sums, op=dist.ReduceOp.SUM))
        elif self.icp_mode == "allreduce":
            c.icp_partition_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"],
                                  self.rank, self.world, self.sums.data_ptr(), lambda: dist.all_reduce(self.sums, op=dist.ReduceOp.SUM))
        else:
            c.icp_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        c.integrate(None, P["integrate_sdf_trunc"], self.integ_dist)
        if self.speculate:
            c.raycast_slab_cross_spec(None, self.inc, P["depth_trunc_min"], self.trunc_max, ex.ta.data_ptr(), self.ta_own.data_ptr(), self.spec.data_ptr())
        else:
            c.raycast_slab_cross(None, self.inc, P["depth_trunc_min"], self.trunc_max, ex.ta.data_ptr())

        if self._merge_events is not None:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            ex.merge()
            e1.record(self.stream)
            self._merge_events.append((e0, e1))
        else:
            ex.merge()

    def time_merge(self, on=True):
        """Measurement legs only: time every frame's merge (both all-reduces, mask, unpack)
        with a torch event pair on the pipeline's stream."""
        self._merge_events = [] if on else None

    def read_merge_ms(self):
        """(total ms, frames) of the merges timed since time_merge(True); resets the list."""
        self.sync()
        ev, self._merge_events = self._merge_events or [], ([] if self._merge_events is not None else None)
        return float(sum(a.elapsed_time(b) for a, b in ev)), len(ev)

    def verify_lockstep(self):
        """Every rank must have fused and lost the same frames and hold the same pose bits -- halo layers are re-integrated by both
        neighbours, so a rank that silently lost a frame (or tracked differently) would let its halo drift from its neighbour's owned
        layers.  A collective (all ranks call it at the same point, outside timed regions); raises on disagreement."""
        import numpy as np
        torch, dist = self.torch, self.dist
        self.sync()
        st = self.stats()
        _, pose, status, _ = self.track_result()
        bits = np.ascontiguousarray(pose, np.float32).view(np.uint32).astype(np.int64).reshape(-1)
        mine = torch.tensor([int(st["frames_fused"]), int(st["frames_lost"]), int(status)] + bits.tolist(), dtype=torch.int64, device=self.ex.ta.device)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        for r, other in enumerate(every):
            if not torch.equal(other, every[0]):
                raise RuntimeError("z-slab ranks out of lock-step: rank 0 (fused, lost, status, pose bits) = %s, rank %d = %s"
                                   % (every[0].tolist()[:3], r, other.tolist()[:3]))
        return dict(frames_fused=int(st["frames_fused"]), frames_lost=int(st["frames_lost"]))

    def sync(self):
        self.ctx.sync()
        self.torch.cuda.synchronize()

    def stats(self, observed=False):
        return self.ctx.stats(observed)

    def stage_timers(self, mask):
        self.ctx.stage_timers(mask)

    def read_stage_ms(self):
        return self.ctx.read_stage_ms()

    def track_result(self):
        return self.ctx.track_result()

    def close(self):
        self.ctx.close()
