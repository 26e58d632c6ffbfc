"""CPU-only: the C-ABI library builds, loads and exports every symbol include/hybkf.h declares (no compute calls)."""
import ctypes as C
import os
import re

from hybkinectfu_amd import lib as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "hybkf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kf_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(K.SYMBOLS)


def test_library_exports_every_symbol():
    K.build()
    lib = K.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.kf_version().decode().startswith("hybkf")
    assert lib.kf_error_string(0) == b"ok"


def test_struct_layouts_match_reference_sizes():
    # src/AppParams.h:36-43 CameraParams 24 B; src/cuda/Mat.h Mat44 64 B; MarchingcubeData.h Triangle 72 B
    assert C.sizeof(K.CameraParams) == 24 and C.sizeof(K.Mat44) == 64 and K.TRI_DTYPE.itemsize == 72
    assert C.sizeof(K.IntegrateParams) == 8 and C.sizeof(K.RaycastParams) == 4


def test_argument_errors_without_gpu():
    lib = K.load()
    assert lib.kf_create(None, None) == 1001
    assert lib.kf_destroy(None) == 1001
    assert lib.kf_synchronize(None) == 1001


def test_product_never_touches_oracle():
    """The shipped path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "hybkinectfu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("the cpu oracle", "").replace("cpu oracle", ""), os.path.join(dirpath, f)


def test_two_virtual_plugin_compiles_and_links(tmp_path):
    """A tracker written against the reference's CameraPoseFinder (src/CameraPoseFinder.h:38-39: two pure virtuals) must build
    against hybkf_host.hpp unchanged and link against libhybkf_host.so."""
    import subprocess
    host = os.path.join(ROOT, "hybkinectfu_amd", "host")
    src = os.path.join(ROOT, "tests", "plugin", "two_virtual_finder.cpp")
    out = str(tmp_path / "libplugin.so")
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-I", host, "-I", os.path.join(ROOT, "include"), src, "-o", out,
                           "-L", os.path.join(ROOT, "hybkinectfu_amd"), "-lhybkf_host", "-lhybkf",
                           "-Wl,-rpath," + os.path.join(ROOT, "hybkinectfu_amd")])
    K.load()
    plug = C.CDLL(out)
    assert plug.plugin_two_virtuals_instantiates() == 0            # a host-side tracker is not device resident


def test_product_library_has_no_experiment_modes():
    """The result-changing timing experiments (KF_*_EXP) exist only in the -DKF_EXPERIMENTS variant (libhybkf_exp.so)."""
    data = open(K.LIB_PATH if not os.environ.get("KF_LIB") else os.path.join(K.PKG_DIR, "libhybkf.so"), "rb").read()
    for name in (b"KF_INTEGRATE_EXP", b"KF_ICP_EXP", b"KF_RAYCAST_EXP"):
        assert name not in data
