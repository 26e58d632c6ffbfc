"""CPU, no reference needed: the packed marching-cubes tables of the HIP kernels and of the oracle hash to the values recorded from the
reference's own header (tests/test_oracle_vs_ref.py::test_marching_cubes_tables_equal_the_reference_header compares element for
element where /root/reference is present and checks the same hashes)."""
import hashlib
import os

import numpy as np

import oracle_lib as O
from test_oracle_vs_ref import MC_EDGE_SHA256, MC_TRI_SHA256, _unpack_mc_inc


def test_mc_tables_hash_without_reference():
    for rel in ("hybkinectfu_amd/csrc/mc_tables.inc", "oracle/mc_tables.inc"):
        e, t = _unpack_mc_inc(os.path.join(O.ROOT, rel))
        assert hashlib.sha256(e.astype(np.int32).tobytes()).hexdigest() == MC_EDGE_SHA256, rel
        assert hashlib.sha256(t.astype(np.int32).tobytes()).hexdigest() == MC_TRI_SHA256, rel
    assert open(os.path.join(O.ROOT, "hybkinectfu_amd/csrc/mc_tables.inc")).read() == open(os.path.join(O.ROOT, "oracle/mc_tables.inc")).read()
