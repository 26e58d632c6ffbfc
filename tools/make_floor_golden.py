#!/usr/bin/env python3
"""Generates tests/golden/floor_*.npz: the oracle's pose sequence over the first frames of a stream and the oracle-vs-perturbed-oracle
divergence floor (tests/tracking_floor.py).  The GPU test (tests/test_gpu_tracking_floor.py) compares the HIP path's poses with
these; tests/test_tracking_floor.py re-derives them on the CPU.   usage: tools/make_floor_golden.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tracking_floor as T
import oracle_lib as O
from hybkinectfu_amd import scene as S

O.set_threads(min(16, os.cpu_count() or 1))
CASES = {
    # BASELINE.json configs[1]: 512^3 @ 4 m, VGA, stock parameters, the first 30 frames of the benchmark stream
    "c2": dict(res=512, size=4.0, cam=S.vga_camera(), n=30, kw={}),
    # the host-class test's configuration (tests/test_gpu_host_classes.py): 128^3 @ 3 m, truncation 5 voxels, 2 m integration gate
    "h128": dict(res=128, size=3.0, cam=S.vga_camera(), n=8, kw=dict(sdf_trunc=5 * 3.0 / 128, integ_dist=2.0)),
    # the same with CameraPoseFinderSDF (tests/test_gpu_host_classes.py::test_hybkinectfu_sdf_tracker_sequence)
    "sdf128": dict(res=128, size=3.0, cam=S.vga_camera(), n=6, kw=dict(sdf_trunc=5 * 3.0 / 128, integ_dist=2.0, sdf_tracker=True)),
}
only = sys.argv[1:]
for name, c in CASES.items():
    if only and name not in only:
        continue
    base, out = T.measure_floor(c["res"], c["size"], c["cam"], c["n"], **c["kw"])
    out.update(res=c["res"], size=c["size"], cam=list(c["cam"]), kw=c["kw"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "floor_%s.npz" % name), poses=base, meta=np.array(json.dumps(out)))
    print(name, json.dumps(out))
