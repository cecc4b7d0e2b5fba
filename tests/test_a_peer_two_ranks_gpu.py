"""GPU test (-m gpu): the peer transport between PROCESSES -- 2 to 4 ranks sharing the one GPU of the box, mailboxes
reached through hipIpcMemHandle, arrival flags raised by another process's kernel (tests/peer_two_ranks_worker.py).
Sorts before the in-process GPU tests: the pytest process must not have touched the GPU when it starts children."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["connect", "mailbox"])
@pytest.mark.parametrize("nx,ny,world,steps,align", [(96, 64, 2, 6, "64"), (64, 96, 2, 5, "none"), (640, 512, 4, 8, "64"),
                                                    (2048, 4096, 2, 6, "64"), (300, 200, 6, 5, "64")])    # 6 ranks: a 2 x 3 mesh, up to five neighbours per rank
def test_peer_transport_between_processes(nx, ny, world, steps, align, mode):
    import torch
    assert not torch.cuda.is_initialized(), "run this file before any in-process GPU test"
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_two_ranks_worker.py"), str(nx),
                                       str(ny), str(steps), align, mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
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
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "ERROR" not in out, out[-3000:]
        assert f"rank {r}: tile" in out and "errors 0 (all ranks 0)" in out, out[-3000:]
