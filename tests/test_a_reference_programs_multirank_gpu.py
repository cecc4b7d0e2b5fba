"""GPU test (-m gpu): the reference's OWN multi-rank test programs -- tests/dist_mem/{test_halos, test_gsum,
test_reduction}.f90, compiled UNMODIFIED against this library's Fortran API layer (tests/Makefile target `dropin`, built
in the container that has /root/reference; the executables travel with the work tree like the library's .so) -- run
with the rank counts and domain sizes of the reference's own Makefile (tests/dist_mem/Makefile:64-80: 10x4/np 2, 4x10/np 2,
10x10/np 4 and 6; gsum 4x10/np 4, 6; reduction 10x10/np 4, 6).

One process per rank, all on the box's one GPU, MAILBOX MODE (DLESM_TRANSPORT=mailbox: RCCL refuses two ranks on one
device, and this mode needs no communication library): the session name travels through the library's file rendezvous as
an RCCL id would, go_decompose / map_comms build the tables, every halo_exchange is stores into the neighbour process's
mailbox through a real IPC mapping, global_sum goes over the host-side board, gather_inner_data copies every rank's block
into the root's buffer through an IPC mapping.  The programs check themselves (the hill() known answers of
test_halos.f90:188, the cell counts of test_gsum.f90:108-110, scatter/gather of test_reduction.f90) and print ERROR
lines on a mismatch.  Sorts before the in-process GPU tests: the pytest process has not touched the GPU when it starts
children."""
import os
import socket
import subprocess
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
DROP = os.path.join(ROOT, "tests", "_dropin")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(exe, world, env_extra, args=()):
    import torch
    assert not torch.cuda.is_initialized(), "run this file before any in-process GPU test"
    path = os.path.join(DROP, exe) if not os.path.isabs(exe) else exe
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (make -C tests dropin, in a container that has /root/reference)")
    port = _free_port()
    # a job token no earlier run can have used: a record left under the same port by a finished job is then never taken for ours
    job = f"pytest-{port}-{os.getpid()}-{time.time_ns()}"
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DLESM_TRANSPORT="mailbox", DLESM_JOB_ID=job,
                   DLESM_BOARD_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        env.pop("DL_ESM_ALIGNMENT", None)
        env.update({k: str(v) for k, v in env_extra.items()})
        procs.append(subprocess.Popen([path, *map(str, args)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} failed:\n{out[-3000:]}"
        assert "ERROR" not in out, f"rank {r} of {world}:\n{out[-3000:]}"
    return outs


@pytest.mark.parametrize("nx,ny,world", [(10, 4, 2), (4, 10, 2), (10, 10, 4), (10, 10, 6)])
@pytest.mark.parametrize("alignment", [None, 8])
def test_reference_test_halos(nx, ny, world, alignment):
    env = {"JPIGLO": nx, "JPJGLO": ny}
    if alignment:
        env["DL_ESM_ALIGNMENT"] = alignment
    outs = _run_ranks("test_halos.exe", world, env)
    assert all("Halo exchange for" in o for o in outs), outs[0][-1500:]


@pytest.mark.parametrize("world", [4, 6])
def test_reference_test_gsum(world):
    outs = _run_ranks("test_gsum.exe", world, {"JPIGLO": 4, "JPJGLO": 10})
    vals = [float(ln.split(":")[-1]) for o in outs for ln in o.splitlines() if "Global sum" in ln]
    assert len(vals) == 4 * world and all(v == 40.0 for v in vals), vals


@pytest.mark.parametrize("world", [4, 6])
def test_reference_test_reduction(world):
    outs = _run_ranks("test_reduction.exe", world, {"JPIGLO": 10, "JPJGLO": 10})
    assert any("Field gathered correctly" in o for o in outs), outs[0][-1500:]
    assert all("Field distributed correctly" in o for o in outs), outs[0][-1500:]


def _checksums(out):
    vals = {}
    for ln in out.splitlines():
        if "checksum" in ln and "=" in ln:
            vals[ln.split("checksum")[0].strip()] = float(ln.split("=")[-1])
    return vals


@pytest.mark.parametrize("fuse", [1, 4])
def test_fortran_jacobi_app_on_four_ranks_equals_one_rank(fuse):
    """examples/jacobi_app.f90 (this repository's GOcean-style application: grid_type, r2d_field, halo_exchange,
    field_checksum + the PSy launch wrappers) on 4 ranks of 300 x 300 in mailbox mode against ONE rank of 600 x 600: the
    same global domain, so the initial and the final checksum agree to the rounding of the rank-order sum.  fuse = 1:
    the time-loop form of the distributed step (frame workgroups store into the neighbours' mailboxes); fuse = 4: four
    time steps per launch with one depth-4 exchange over the mailboxes."""
    exe = os.path.join(ROOT, "dl_esm_inf_amd", "fortran", "build", "jacobi_app.exe")
    four = _run_ranks(exe, 4, {"DL_ESM_ALIGNMENT": 64}, args=(300, 24, fuse, 0))
    one = _run_ranks(exe, 1, {"DL_ESM_ALIGNMENT": 64}, args=(600, 24, fuse, 0))
    a, b = _checksums(four[0]), _checksums(one[0])
    assert set(a) == set(b) == {"initial", "final"}, (four[0][-1500:], one[0][-1500:])
    for k in a:
        assert abs(a[k] - b[k]) <= 1e-12 * abs(b[k]), (k, a[k], b[k])
    assert a["final"] != a["initial"]


def _build_c_demo(tmp_path):
    exe = str(tmp_path / "mailbox_demo")
    libdir = os.path.join(ROOT, "dl_esm_inf_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "examples", "mailbox_demo.c"),
                           "-L" + libdir, "-ldlesm_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("nx,ny", [(600, 600), (1000, 260)])
def test_plain_c_job_without_a_communication_library(tmp_path, nx, ny):
    """examples/mailbox_demo.c -- plain C99 over the C ABI, nothing else in the process: the same global domain on 1, 2, 4
    and 6 ranks (mailbox mode: session name through the file rendezvous, plans that connect their own mailboxes, the
    time-loop form of the distributed step, global sums over the board) ends in the same checksums"""
    exe = _build_c_demo(tmp_path)
    got = {}
    for world in (1, 2, 4, 6):
        outs = _run_ranks(exe, world, {"DL_ESM_ALIGNMENT": 8}, args=(nx, ny, 24))
        line = [ln for ln in outs[0].splitlines() if ln.startswith("G: checksum")]
        assert len(line) == 1 and all("G: checksum" not in o for o in outs[1:]), outs[0][-1500:]
        got[world] = [float(v) for v in line[0].split()[2:]]
        assert f"G: ranks {world} " in outs[0] and f"mailbox {1 if world > 1 else 0}" in outs[0], outs[0][-800:]
    for world in (2, 4, 6):
        for a, b in zip(got[world], got[1]):
            assert abs(a - b) <= 1e-12 * abs(b), (world, got[world], got[1])
    assert got[1][0] != got[1][1]
