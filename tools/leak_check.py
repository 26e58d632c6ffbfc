import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hybkinectfu_amd import lib as K, scene as S
P = S.STOCK
cam = S.vga_camera()
def cycle(n, work):
    torch.cuda.synchronize()
    f0, _ = torch.cuda.mem_get_info()
    for i in range(n):
        c = K.Context(K.camera(*cam), 256, 3.0, P["volume_max_weight"], levels=3)
        if work: work(c)
        c.close()
    torch.cuda.synchronize()
    f1, _ = torch.cuda.mem_get_info()
    return (f0 - f1) / 2**20 / n
mm = S.render_depth_mm(S.trajectory_pose(0, 3.0), cam, 3.0)
dev = torch.from_numpy(mm.astype(np.int16)).cuda()
print("create/close only: %.2f MB per cycle" % cycle(20, None))
print("create/close only (again): %.2f MB per cycle" % cycle(20, None))
print("+ upload (copy stream): %.2f MB per cycle" % cycle(20, lambda c: c.upload_depth_mm(mm)))
def pf(c):
    c.set_depth_mm_device(dev.data_ptr()); c.preprocess(0.3, 3.5, 2.0, 0.03)
    c.prefetch_frame(dev.data_ptr(), 0.3, 3.5, 2.0, 0.03)
print("+ prefetch (side stream): %.2f MB per cycle" % cycle(20, pf))
def full(c):
    c.set_pose(S.pose0(3.0))
    for k in range(3):
        c.upload_depth_mm(mm)
        c.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
        c.icp_track(k, P["icp_thre_dist"], P["icp_thre_sin_angle"], P["camera_shake_dist"], P["camera_shake_angle"])
        c.integrate(None, 0.06, 2.5)
        c.raycast(None, 0.04, P["depth_trunc_min"], P["depth_trunc_max"])
    c.track_result()
print("+ three full frames: %.2f MB per cycle" % cycle(20, full))
print("+ three full frames (again: one-time allocations of the runtime are behind us): %.2f MB per cycle" % cycle(40, full))
def col(n):
    torch.cuda.synchronize(); f0, _ = torch.cuda.mem_get_info()
    for i in range(n):
        c = K.Context(K.camera(*cam), 256, 3.0, P["volume_max_weight"], levels=3, max_triangles=100000, has_color=True)
        c.upload_rgb(np.zeros((cam[1], cam[0], 3), np.uint8)); c.close()
    torch.cuda.synchronize(); f1, _ = torch.cuda.mem_get_info()
    return (f0 - f1) / 2**20 / n
print("colour context + rgb upload + triangles: %.2f MB per cycle" % col(20))
