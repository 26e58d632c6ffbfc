import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # every Context.stats() call of the test-suite also sweeps the volume (kf_count_observed_voxels) and compares it with the running count that
    # kf_get_volume_stats reports: each weight_gt0 assertion against the oracle thereby checks both
    os.environ.setdefault("KF_STATS_CROSSCHECK", "1")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# environment knobs that change WHICH tracking launch form runs (not its results): tests that assert a launch form only do so when none is set,
# so that `KF_ICP_PERSISTENT=0 pytest -m gpu` and the like still check every result against the oracle
FORM_KNOBS = ("KF_ICP_PERSISTENT", "KF_SDF_PERSISTENT", "KF_ICP_BATCHED", "KF_ICP_COOPERATIVE", "KF_CULL_IN_TRACK", "KF_PREFETCH_IN_TRACK",
              "KF_SDF_LOOP_WG")


def default_forms():
    return not any(k in os.environ for k in FORM_KNOBS)


needs_default_forms = pytest.mark.skipif(not default_forms(), reason="asserts the default launch forms; a KF_* form knob is set")


@pytest.fixture(autouse=True)
def _close_leaked_contexts():
    """A test that fails half-way leaves its kf_ctx alive in the traceback; a second live context switches the persistent loops off and
    would fail every later form assertion.  Close whatever is left after each test."""
    yield
    try:
        from hybkinectfu_amd import lib as K
    except Exception:
        return
    for c in K.live_contexts():
        c.close()
