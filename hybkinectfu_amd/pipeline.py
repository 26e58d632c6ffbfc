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

    def process_frame_device(self, dev_mm_ptr, frame_id):
        c = self.ctx
        c.set_depth_mm_device(dev_mm_ptr)                                                                    # copyFrameToGPU
        c.preprocess(P["depth_trunc_min"], self.trunc_max, P["filter_sigma_pixel"], P["filter_sigma_depth"])  # :106-110
        c.icp_track(frame_id, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])  # :116
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
