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
