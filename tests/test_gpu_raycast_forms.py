"""GPU: the raycast's switchable forms give the same bits.

The crossing's gradient (gradientForPoint, src/cuda/raycastingVolume.cu:16-42) is evaluated from one shared 32-voxel neighbourhood
(csrc/grad_shared.h) with a wave-level fallback to six separate lookups for taps whose cell is not "the vertex's cell moved by one".  Real data never
takes the fallback, so KF_RAYCAST_SHARED_GRAD=2 sends every other wave down it; 0 switches the shared form off.  The maps of all three -- whole
volume (k_raycast) and a stored z-slab (k_slab_ray_normals) -- and of the launches without tile bounds / meso table must be identical.  That the
default form equals the ORACLE bit for bit is test_gpu_parity.py's business (same process, default switches)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _digest(env_extra, res, cols, rows):
    env = dict(os.environ)
    for k in ("KF_RAYCAST_SHARED_GRAD", "KF_RAYCAST_VIEW_HALF", "KF_RAYCAST_BOUNDS", "KF_RAYCAST_MESO"):
        env.pop(k, None)
    env.update(env_extra)
    env["PYTHONPATH"] = os.path.dirname(HERE) + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(HERE, "raycast_digest.py"), str(res), str(cols), str(rows)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("res,cols,rows", [(128, 320, 240), (96, 200, 150)])
def test_gradient_forms_and_table_switches_give_the_same_maps(res, cols, rows):
    ref = _digest({}, res, cols, rows)
    assert ref["whole0hits"] > cols * rows // 3 and ref["whole1hits"] > cols * rows // 3 and ref["slab0hits"] > 200       # there is something to compare
    # (KF_RAYCAST_VIEW_HALF: the gathers' raw-buffer view reaches that many brick layers to either side of the wave's first lane instead of 2 GB worth --
    # a small volume then meets what 1024^3 / 2048^3 meet: views that start inside the volume, waves whose lanes do not fit one view)
    for env in ({"KF_RAYCAST_SHARED_GRAD": "0"}, {"KF_RAYCAST_SHARED_GRAD": "2"}, {"KF_RAYCAST_VIEW_HALF": "1"}, {"KF_RAYCAST_VIEW_HALF": "3"},
                {"KF_RAYCAST_BOUNDS": "0", "KF_RAYCAST_MESO": "0"}):
        assert _digest(env, res, cols, rows) == ref, env
