"""GPU test (-m gpu): the comparison-only forms -- y-march and LDS-staged Jacobi sweeps, other tile heights, shuffles
instead of DPP wave shifts, stacked / straight-line / old-level-first shallow-water tiles, the pipeline form of the fused
steps (LAB_NOTES.md) -- are not in the product library.  They are compiled from the SAME sources into libdlesm_hip_lab.so
(-DDLESM_LAB) and every test case that selects one runs HERE: one child pytest with DLESM_HIP_LIB pointing at the lab
build, in which tests/conftest.py keeps exactly the cases this process deselects.  The parity results of the main run are
therefore the product library's, and no variant loses its coverage.

Sorts before the in-process GPU tests: the pytest process must not have touched the GPU when it starts a child."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT
from dl_esm_inf_amd import _cabi

pytestmark = pytest.mark.gpu


def test_comparison_only_forms_in_the_lab_build():
    import torch
    assert not torch.cuda.is_initialized(), "run this file before any in-process GPU test"
    assert os.path.exists(_cabi.LAB_BUILD_PATH), f"{_cabi.LAB_BUILD_PATH} not built (make -C dl_esm_inf_amd/csrc lab)"
    env = dict(os.environ, DLESM_HIP_LIB=_cabi.LAB_BUILD_PATH)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-q", "-x", "--timeout=600",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail + p.stderr[-2000:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 700, tail        # the variants of the Jacobi, fused-step and shallow-water sweeps
    assert "failed" not in tail.splitlines()[-1], tail


def test_the_product_library_does_not_hold_them():
    """a lab key reads as its default in the product library (set: ignored, said once on stderr), and the library says
    which build it is"""
    import torch
    assert not torch.cuda.is_initialized(), "run this file before any in-process GPU test"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dl_esm_inf_amd import _cabi\n"
            "L = _cabi.lib()\n"
            "print('lab', L.dlesm_is_lab_build(), L.dlesm_tuning_class(b'j5_kernel'), L.dlesm_tuning_class(b'dm_safe'), "
            "L.dlesm_tuning_class(b'sw_kernel'), L.dlesm_tuning_class(b'no_such_key'))\n"
            "L.dlesm_set_tuning(b'j5_kernel', 1); L.dlesm_set_tuning(b'j5_kernel', 2)\n" % ROOT)
    for lib, want in ((_cabi.LIB_PATH, "lab 0 2 0 1 -1"), (_cabi.LAB_BUILD_PATH, "lab 1 2 0 1 -1")):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DLESM_HIP_LIB=lib), capture_output=True, text=True,
                           timeout=300)
        assert p.returncode == 0 and want in p.stdout, p.stdout + p.stderr
        said = p.stderr.count("exists in libdlesm_hip_lab.so only")
        assert said == (1 if lib == _cabi.LIB_PATH else 0), p.stderr
