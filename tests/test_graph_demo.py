"""examples/graph_demo.c: a distributed time loop captured into ONE hipGraph from plain C (system HIP
runtime and RCCL, no torch in the process).  CPU: it builds and stops at the documented status without
a device.  GPU (-m gpu): stepwise == replayed graph, bit for bit, and both equal the oracle."""
import ctypes as C
import os
import subprocess
import sys
import types

import pytest

import oracle_lib as O
from conftest import ROOT

LIBDIR = os.path.join(ROOT, "dl_esm_inf_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "graph_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "graph_demo.c"), "-L" + LIBDIR, "-ldlesm_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    return exe


def _env():
    e = dict(os.environ)
    e.pop("DL_ESM_ALIGNMENT", None)
    e.pop("LD_LIBRARY_PATH", None)          # the system ROCm, as a Fortran or C host program would get
    return e


def test_graph_demo_builds_and_refuses_to_run_without_a_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu test")
    p = subprocess.run([exe, "64", "48", "4"], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "no HIP device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl", "peer"])
@pytest.mark.parametrize("nx,ny,nsteps", [(500, 300, 8), (130, 7, 6)])
def test_graph_replay_equals_stepwise_and_the_oracle(tmp_path, nx, ny, nsteps, transport):
    """transport = peer: the plan's mailboxes connected first -- the captured steps hold no RCCL call, their sequence numbers
    advance on the device from replay to replay"""
    exe = _build(tmp_path)
    p = subprocess.run([exe, str(nx), str(ny), str(nsteps)] + (["peer"] if transport == "peer" else []), env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    g = {ln.split()[1]: ln.split()[2:] for ln in p.stdout.splitlines() if ln.startswith("G: ")}
    assert g["transport"] == ["mailboxes" if transport == "peer" else "rccl"]
    assert g["stepwise"] == g["graph"], p.stdout          # the printed 17 digits: bit for bit
    ld, nyarr = O.grid_extents(nx + 2, ny + 2)
    assert [int(v) for v in g["grid"]] == [ld, nyarr]
    # oracle: the same loop-back tables, hash on the internal box, full exchange, then edges-only steps
    import dl_esm_inf_amd as D
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from dm_overhead import loopback_tables
    it = types.SimpleNamespace(xstart=2, xstop=nx + 1, ystart=2, ystop=ny + 1, nx=nx, ny=ny)
    t = loopback_tables(D, it)
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    a = O.hash_field(20261004, nyarr, ld, 0, 0, 2, nx + 1, 2, ny + 1)
    assert O.exchange_all([a], [ld], [oc]) == 0
    b = a.copy()
    for _ in range(nsteps):
        O.jacobi5(a, b, ld, 2, nx + 1, 2, ny + 1)
        assert O.exchange_dirs([b], [ld], [oc], (1, 2, 3, 4), no_diagonals=True) == 0
        a, b = b, a
    cs = O.lib().orc_checksum(a, ld, 1, nx + 2, 1, ny + 2)
    assert abs(float(g["graph"][0]) - cs) <= 1e-12 * cs


def test_mailbox_demo_builds_and_refuses_to_run_without_a_device(tmp_path):
    """examples/mailbox_demo.c (a multi-rank job from plain C with no communication library): strict C99 build against the
    header; without a device it stops with the documented status (the multi-rank runs are in
    tests/test_a_reference_programs_multirank_gpu.py)"""
    import torch
    exe = str(tmp_path / "mailbox_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "examples", "mailbox_demo.c"),
                           "-L" + LIBDIR, "-ldlesm_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + LIBDIR,
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu test")
    p = subprocess.run([exe, "64", "48", "4"], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "no HIP device" in p.stderr
