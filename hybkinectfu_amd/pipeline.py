"""Per-frame pipelines over the C ABI, mirroring HybKinectfu::processNewFrame (src/HybKinectfu.cpp:98-160).

SingleGpuPipeline: one context, the whole volume.  Everything a frame needs is enqueued on the context's stream with
no host synchronisation: the pose and the tracking verdict stay on the device (kf_icp_track), integrate is predicated on
the verdict inside the kernel (src/HybKinectfu.cpp:123), and raycast reads the device-resident pose.
"""
from . import lib as K
from . import scene as S

P = S.STOCK


class SingleGpuPipeline:
    def __init__(self, kcam, res, size, wl=None, device=0, max_triangles=0):
        wl = wl or {}
        self.cam = kcam
        self.trunc_max = wl.get("trunc_max", P["depth_trunc_max"])
        self.integ_dist = wl.get("integ_dist", P["integrate_depth_trunc"])
        self.ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=max_triangles, device=device)
        self.ctx.set_pose(S.pose0(size))                       # HybKinectfu::init  src/HybKinectfu.cpp:51-57
        self.inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]     # AppParamsProducer.cpp:113-117

    def process_frame_device(self, dev_mm_ptr, frame_id, next_mm_ptr=None):
        """next_mm_ptr: where the NEXT frame already lies in HBM (streaming input): its preprocess is enqueued on the context's
        side stream right behind the tracking loop and overlaps with it."""
        c = self.ctx
        c.set_depth_mm_device(dev_mm_ptr)                                                                    # copyFrameToGPU
        c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])  # :106-110
        c.icp_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])  # :116
        if next_mm_ptr is not None:
            c.prefetch_frame(next_mm_ptr, P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        c.integrate(None, P["integrate_sdf_trunc"], self.integ_dist)                                         # :125-140
        c.raycast(None, self.inc, P["depth_trunc_min"], self.trunc_max)                                      # :149-154

    def sync(self):
        self.ctx.sync()

    def stats(self):
        return self.ctx.stats()

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
SLAB_HALO = 16      # voxel layers stored AND integrated on each side of the owned range: covers the previous ray sample
                    # (ray increment <= ~6 voxels at 1024^3 @ 6 m), the 2x2x2 trilinear taps and the +-1-cell gradient taps


def slab_ranges(res, world):
    """Owned z range [z0, z1) of every rank: contiguous, brick (8-layer) aligned, sizes differing by at most one brick."""
    nb = res // 8
    base, extra = divmod(nb, world)
    out, b = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((b * 8, (b + n) * 8))
        b += n
    return out


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


class SlabPipeline:
    """One rank of the z-slab partitioned pipeline.

    Replicated per rank (cheap, and bitwise identical everywhere): preprocess, pyramids, the whole ICP loop.
    Partitioned: TSDF integrate (own slab + halo, no communication) and raycast (own samples only), followed by one
    MIN all-reduce of the crossing parameter and one integer SUM all-reduce of the winning vertex/normal maps.
    """

    def __init__(self, kcam, res, size, wl=None, rank=0, world=1, device=0, max_triangles=0, icp_mode="replicated"):
        import torch
        import torch.distributed as dist
        self.icp_mode = icp_mode            # "replicated" (default, faster at VGA) or "allreduce" (pixels split over the ranks)
        wl = wl or {}
        self.dist, self.torch = dist, torch
        self.rank, self.world = rank, world
        self.trunc_max = wl.get("trunc_max", P["depth_trunc_max"])
        self.integ_dist = wl.get("integ_dist", P["integrate_depth_trunc"])
        self.inc = P["raycast_increment_factor"] * P["integrate_sdf_trunc"]
        self.slab = slab_ranges(res, world)[rank]
        self.ctx = K.Context(kcam, res, size, P["volume_max_weight"], levels=3, max_triangles=max_triangles, device=device,
                             slab=self.slab, halo=SLAB_HALO)
        self.ctx.set_pose(S.pose0(size))
        dev = torch.device("cuda", device)
        # ONE stream orders the library's kernels, torch's elementwise ops and the RCCL collectives (which synchronise with the
        # CURRENT torch stream).  It has to be a stream of its own: torch's default stream has the null handle, which
        # kf_set_stream reads as "back to the private stream" -- the kernels would then race with the collectives.
        self.stream = torch.cuda.Stream(device=dev)
        assert self.stream.cuda_stream != 0
        self.ctx.set_stream(self.stream.cuda_stream)
        self.t = torch.empty((kcam.rows, kcam.cols), dtype=torch.float32, device=dev)
        self.tmin = torch.empty_like(self.t)
        # vertex and normal candidates share one buffer, so their merge is ONE integer SUM all-reduce (9.8 MB at VGA)
        self.vn = torch.empty((2, kcam.rows, kcam.cols, 4), dtype=torch.float32, device=dev)
        self.v, self.n = self.vn[0], self.vn[1]
        self.vn_bits = self.vn.view(torch.int32)
        # what actually crosses xGMI: vertex xyz + normal xyz of the winner, 24 bytes per pixel (7.4 MB at VGA)
        self.packed = torch.empty((kcam.rows, kcam.cols, 6), dtype=torch.float32, device=dev)
        self.packed_bits = self.packed.view(torch.int32)
        self.sums = torch.zeros(32, dtype=torch.float32, device=dev)
        self._preprocessed = None           # device pointer of a frame whose preprocess was enqueued during the previous frame's merge

    def process_frame_device(self, dev_mm_ptr, frame_id, next_mm_ptr=None):
        with self.torch.cuda.stream(self.stream):
            self._process_frame_device(dev_mm_ptr, frame_id, next_mm_ptr)

    def _process_frame_device(self, dev_mm_ptr, frame_id, next_mm_ptr):
        c, dist = self.ctx, self.dist
        if self._preprocessed != dev_mm_ptr:                              # not done ahead of time by the previous call (see below)
            c.set_depth_mm_device(dev_mm_ptr)
            c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
        self._preprocessed = None
        if self.icp_mode == "allreduce":
            c.icp_partition_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"],
                                  self.rank, self.world, self.sums.data_ptr(), lambda: dist.all_reduce(self.sums, op=dist.ReduceOp.SUM))
        else:
            c.icp_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        c.integrate(None, P["integrate_sdf_trunc"], self.integ_dist)
        c.raycast_slab(None, self.inc, P["depth_trunc_min"], self.trunc_max, self.t.data_ptr(), self.v.data_ptr(), self.n.data_ptr())
        # first crossing along each ray wins (merge_candidates above is the same rule in plain torch, used by the CPU tests):
        # MIN all-reduce of t, mask the losers and pack xyz + xyz on the device, ONE integer SUM all-reduce, unpack into the maps
        self.tmin.copy_(self.t)
        pending = dist.all_reduce(self.tmin, op=dist.ReduceOp.MIN, async_op=True)
        if next_mm_ptr is not None:
            # the collective runs on RCCL's own stream until wait() joins it: the next frame's preprocess -- every reader of this
            # frame's maps is already behind us in the stream -- fills that time instead of the start of the next frame
            c.set_depth_mm_device(next_mm_ptr)
            c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])
            self._preprocessed = next_mm_ptr
        pending.wait()
        c.slab_pack_candidates(self.t.data_ptr(), self.tmin.data_ptr(), self.v.data_ptr(), self.n.data_ptr(), self.packed.data_ptr())
        dist.all_reduce(self.packed_bits, op=dist.ReduceOp.SUM)
        c.set_model_maps_packed(self.packed.data_ptr())

    def sync(self):
        self.ctx.sync()
        self.torch.cuda.synchronize()

    def stats(self):
        return self.ctx.stats()

    def stage_timers(self, mask):
        self.ctx.stage_timers(mask)

    def read_stage_ms(self):
        return self.ctx.read_stage_ms()

    def track_result(self):
        return self.ctx.track_result()

    def close(self):
        self.ctx.close()
