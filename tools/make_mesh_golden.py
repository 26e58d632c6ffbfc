#!/usr/bin/env python3
"""Generate tests/golden/mesh_*.npz: what the REFERENCE's own host mesh code makes of a triangle soup.

Runs only in the build container: the expected outputs come from oracle/_ref/libkfrefmesh.so = /root/reference/src's
ml::MeshData<float> + ml::MeshIO<float> compiled as they lie (oracle/ref_mesh_harness.cpp replays
MeshGeneratorMarchingcube::saveMesh, src/MeshGeneratorMarchingcube.cpp:61-96).  Each fixture holds the input soup and the
reference's welded vertices / faces / normals / colours plus the bytes of the OBJ, PLY and OFF files it wrote.
Cases: the marching-cubes soups of the s32 / s64 golden volumes (tests/golden/s*.npz: tri_pos, the GPU's marching cubes
reproduces them bit for bit), the s64 soup with per-vertex colours, and a synthetic stress soup (clusters closer than the
1e-4 m weld cell, negative coordinates, duplicated and collapsing triangles).   usage: python tools/make_mesh_golden.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_mesh  # noqa: E402
from hybkinectfu_amd import lib as K  # noqa: E402


def soup_from_positions(pos, color=None):
    tris = np.zeros(len(pos), dtype=K.TRI_DTYPE)
    tris["v"]["pos"] = pos
    if color is not None:
        tris["v"]["color"] = color
    return tris


def stress_soup(seed=7, n=3000):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-0.02, 0.02, size=(400, 3)).astype(np.float32)          # both signs: sign(v)*0.5 in toVirtualVoxelPos
    idx = rng.integers(0, len(centres), size=(n, 3))
    jitter = rng.choice(np.array([0.0, 2e-5, -3e-5, 6e-5, 1.4e-4], np.float32), size=(n, 3, 3))
    pos = centres[idx] + jitter                                                   # many vertices within one or two weld cells
    pos[::17] = pos[1::17][: len(pos[::17])]                                      # exact duplicate triangles
    pos[5::23, 1] = pos[5::23, 0] + np.float32(1e-5)                              # an edge the weld collapses -> degenerate face
    pos[3] = 0.0                                                                  # sign(0) = 0
    return pos.astype(np.float32)


def make(name, tris, with_color):
    with tempfile.TemporaryDirectory() as d:
        cwd = os.getcwd()
        os.chdir(d)                                       # the OBJ header quotes the file name: keep it relative and fixed
        try:
            m = ref_mesh.process(tris, with_color, save_as=("mesh.obj", "mesh.ply", "mesh.off"))
            files = {ext: np.frombuffer(open("mesh." + ext, "rb").read(), np.uint8) for ext in ("obj", "ply", "off")}
        finally:
            os.chdir(cwd)
    out = dict(soup=tris.view(np.float32).reshape(len(tris), 18), with_color=np.array([int(with_color)]), vertices=m["vertices"],
               normals=m["normals"], colors=m["colors"], faces=m["faces"], obj=files["obj"], ply=files["ply"], off=files["off"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print("%s: %d triangles -> %d vertices, %d faces (reference MeshData); obj %d B, ply %d B, off %d B" %
          (name, len(tris), len(m["vertices"]), len(m["faces"]), len(files["obj"]), len(files["ply"]), len(files["off"])))


if __name__ == "__main__":
    assert ref_mesh.available(), "oracle/_ref/libkfrefmesh.so missing: `make -C oracle ref` (needs /root/reference)"
    g32 = np.load(os.path.join(ROOT, "tests", "golden", "s32.npz"))
    g64 = np.load(os.path.join(ROOT, "tests", "golden", "s64.npz"))
    make("mesh_s32", soup_from_positions(g32["tri_pos"]), False)
    make("mesh_s64", soup_from_positions(g64["tri_pos"]), False)
    p = g64["tri_pos"]
    col = (np.floor(np.abs(p) * 997.0) % 256 / 255.0).astype(np.float32)           # per-vertex b,g,r in [0,1], a function of position
    make("mesh_s64_color", soup_from_positions(p, col), True)
    make("mesh_stress", soup_from_positions(stress_soup()), False)
