import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: exercises a comparison-only form that exists in libdlesm_hip_lab.so only")


# ---- product library vs lab build -------------------------------------------------------------------------------------
# libdlesm_hip.so (the product) holds the forms it can reach by itself; the comparison-only kernels (y-march and LDS-staged
# Jacobi sweeps, other tile heights, shuffles instead of DPP, stacked / straight-line shallow-water tiles, the pipeline form
# of the fused steps) are compiled into libdlesm_hip_lab.so only (-DDLESM_LAB, the same sources).  A test case that selects
# one of them runs against the lab build, in the ONE child pytest that tests/test_a_lab_build_gpu.py starts with
# DLESM_HIP_LIB pointing at it; this process -- whose parity results are the product's -- deselects those cases, and the
# child deselects everything else.  The rule below mirrors the library's own classes (dlesm_tuning_class; checked by
# tests/test_cabi_host.py::test_tuning_keys_are_classified).
LAB_RUN = os.path.basename(os.environ.get("DLESM_HIP_LIB", "")) == "libdlesm_hip_lab.so"
_LAB_DEFAULTS = {"j5_kernel": 0, "j5_rows": 0, "j5_unroll": 4, "j5xt_rows": 0, "j5xt_dpp": 1, "j5xt_march": 0,
                 "j5xt_march_ring": 9, "j5xt_march_slots": 3072, "j5xt_march_perm": 1, "sw_tile_rows": 2, "sw_dpp": 1,
                 "sw_stack": 1, "sw_dm_diag": 0, "dm_event_system_fence": 0, "sw_x2_rows": 0, "sw_x2_nt": 2, "sw_x2_stack": 4,
                 "sw_x2_pad": 0, "sw_x2_sw_form": 0}
_PARAM_KEYS = {"sw_rows": "sw_tile_rows", "sw_dpp": "sw_dpp", "sw_nt": "sw_nt", "march": "j5xt_march", "stack": "sw_stack", "x2_rows": "sw_x2_rows"}


def needs_lab(tune):
    """does this set of tuning values select a form the product library does not hold?"""
    for k, v in tune.items():
        if k in _LAB_DEFAULTS and v != _LAB_DEFAULTS[k]:
            return True
    if tune.get("j5_tile_rows", 0) not in (0, 2, 3) or (tune.get("j5_variant", 0) & 0x0B) or tune.get("sw_nt", 2) >= 4:
        return True
    return False


def _item_needs_lab(item):
    if item.get_closest_marker("lab") is not None:
        return True
    cs = getattr(item, "callspec", None)
    if cs is None:
        return False
    tune = {}
    for name, val in cs.params.items():
        if isinstance(val, dict):
            tune.update({k: v for k, v in val.items() if isinstance(v, int)})
        elif name in _PARAM_KEYS and isinstance(val, int):
            tune[_PARAM_KEYS[name]] = val
    return needs_lab(tune)


def pytest_collection_modifyitems(config, items):
    keep, drop = [], []
    for it in items:
        (keep if _item_needs_lab(it) == LAB_RUN else drop).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden
