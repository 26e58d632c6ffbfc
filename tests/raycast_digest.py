"""Child process of test_gpu_raycast_forms.py: fuse a few frames of Scene S, raycast from two poses (whole volume; then a context that stores only a
z-slab, whose crossing words + owner normals go through the slab kernels) and print one SHA-1 per map.  The environment switches of the raycast
(KF_RAYCAST_SHARED_GRAD, KF_RAYCAST_BOUNDS, KF_RAYCAST_MESO) are read once per process -- hence a process per setting."""
import hashlib
import json
import sys

import numpy as np
import torch

torch.zeros(1, device="cuda:0")          # (torch's HIP runtime first, as everywhere in the suite: initialised after the library's, it finds no device)

from hybkinectfu_amd import lib as K
from hybkinectfu_amd import scene as S

P = S.STOCK


def main(res, cols, rows):
    size = 3.0
    cam = (cols, rows, (cols - 1) / 2.0, (rows - 1) / 2.0, 525.0 * cols / 640.0, 525.0 * cols / 640.0)
    trunc = 5 * size / res
    out = {}
    for name, slab in (("whole", None), ("slab", (res // 4, res // 2 + 8))):
        ctx = K.Context(K.camera(*cam), res, size, P["volume_max_weight"], levels=3, slab=slab, halo=16 if slab else 0)
        for k in range(3):
            pose = S.trajectory_pose(3 * k, size).astype(np.float32)
            ctx.upload_depth_mm(S.render_depth_mm(pose, cam, size))
            ctx.preprocess(P["depth_trunc_min"], P["depth_trunc_max"], P["filter_sigma_pixel"], P["filter_sigma_depth"])
            ctx.integrate(pose, trunc, 2.5)
        for j, k in enumerate((6, 11)):
            pose = S.trajectory_pose(k, size).astype(np.float32)
            if slab is None:
                ctx.raycast(pose, 0.7 * trunc, P["depth_trunc_min"], P["depth_trunc_max"])
                for m, tag in ((K.MAP_MODEL_VERTICES, "v"), (K.MAP_MODEL_NORMALS, "n")):
                    a = ctx.download_map(m)
                    out["%s%d%s" % (name, j, tag)] = hashlib.sha1(np.ascontiguousarray(a).view(np.uint8)).hexdigest()
                    if tag == "n":
                        out["%s%dhits" % (name, j)] = int((np.abs(a[..., :3]).sum(axis=-1) > 0).sum())
            else:
                ta = torch.empty((rows, cols), dtype=torch.int64, device="cuda:0")
                cand = torch.empty((rows, cols, 3), dtype=torch.float32, device="cuda:0")
                ctx.raycast_slab_cross(pose, 0.7 * trunc, P["depth_trunc_min"], P["depth_trunc_max"], ta.data_ptr())
                ctx.slab_ray_normals(pose, 0.7 * trunc, P["depth_trunc_min"], P["depth_trunc_max"], ta.data_ptr(), cand.data_ptr())
                ctx.sync()
                ta, cand = ta.cpu().numpy(), cand.cpu().numpy()
                out["%s%dta" % (name, j)] = hashlib.sha1(np.ascontiguousarray(ta).view(np.uint8)).hexdigest()
                out["%s%dcand" % (name, j)] = hashlib.sha1(np.ascontiguousarray(cand).view(np.uint8)).hexdigest()
                out["%s%dhits" % (name, j)] = int((cand.view(np.uint32) != 0).any(axis=-1).sum())
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:4]))
