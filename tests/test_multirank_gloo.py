"""N>1 path on CPU: real processes, gloo backend (the reference's own multi-rank strategy is
real MPI processes on one box, tests/dist_mem/Makefile:64-80).  Each rank runs
tests/gloo_worker.py, which takes its tile and message tables from the product's C ABI and
plays test_halos / test_gsum / test_reduction on them."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, args):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        env.pop("DL_ESM_ALIGNMENT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_worker.py")] +
                                      [str(a) for a in args], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "ERROR" not in out


# the depth-d extension: deep-halo exchange + the staged steps of the fused distributed kernel
@pytest.mark.parametrize("nx,ny,world,depth", [(12, 8, 2, 2), (8, 13, 2, 3), (17, 16, 4, 2), (24, 21, 6, 4),
                                               (16, 32, 8, 4), (32, 33, 4, 8), (20, 40, 2, 6)])
def test_deep_halos_and_staged_steps_over_gloo(nx, ny, world, depth):
    _run_ranks(world, [nx, ny, depth])


# world_size 2 (x-split and y-split) as the contract asks, plus the reference's 4- and 6-rank cases
# ... and BASELINE configs[4]'s 2 x 4 mesh (16384 x 32768 over 8 ranks) at a small tile, uneven 3 x 3 and 1 x 5 meshes
@pytest.mark.parametrize("nx,ny,world", [(10, 4, 2), (4, 10, 2), (10, 10, 4), (10, 10, 6), (16, 32, 8), (13, 11, 9),
                                         (7, 40, 5)])
def test_dist_mem_suite_over_gloo(nx, ny, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        env.pop("DL_ESM_ALIGNMENT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_worker.py"),
                                       str(nx), str(ny)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "ERROR" not in out
