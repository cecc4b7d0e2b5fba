"""examples/c_abi_demo.c: the C ABI bound from plain C99 (the header compiles as C, the library links
without Python, Fortran or torch).  CPU: build it, run the host-side part (no device: it must stop with
the documented status, not fall back).  GPU (-m gpu): run it and compare with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT

LIBDIR = os.path.join(ROOT, "dl_esm_inf_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + LIBDIR, "-ldlesm_hip",
                           "-L/opt/rocm/lib", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    return exe


def _env():
    e = dict(os.environ)
    e.pop("DL_ESM_ALIGNMENT", None)
    return e


def test_c_program_builds_and_refuses_to_run_without_a_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu test")
    p = subprocess.run([exe, "64", "48", "3"], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "no HIP device" in p.stderr
    ld, ny = O.grid_extents(66, 50)
    assert p.stdout.split()[:8] == ["G:", "grid", str(ld), str(ny), "internal", "2", "65", "2"]


@pytest.mark.gpu
def test_c_program_matches_the_oracle_on_gpu(tmp_path):
    exe = _build(tmp_path)
    nx, ny, nsteps = 1000, 600, 10
    p = subprocess.run([exe, str(nx), str(ny), str(nsteps)], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    g = {ln.split()[1]: ln.split()[2:] for ln in p.stdout.splitlines() if ln.startswith("G: ")}
    ld, nyarr = O.grid_extents(nx + 2, ny + 2)
    assert [int(x) for x in g["grid"][:2]] == [ld, nyarr]
    a = O.hash_field(20261004, nyarr, ld, 0, 0, 1, nx + 2, 1, ny + 2)
    b = a.copy()
    for _ in range(nsteps):
        O.jacobi5(a, b, ld, 2, nx + 1, 2, ny + 1)
        a, b = b, a
    cs = O.lib().orc_checksum(a, ld, 2, nx + 1, 2, ny + 1)
    assert abs(float(g["checksum"][0]) - cs) <= 1e-12 * cs
    assert [float(x) for x in g["patch"]] == [a[1, 1], a[1, 3], a[2, 1], a[2, 3]]
