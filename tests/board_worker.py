"""One process of the CPU test of the BOARD (dlesm_board_*: the host-side all-gather of mailbox mode, no GPU, no RCCL):
    python tests/board_worker.py SESSION RANK NRANKS DIR"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
session, rank, n, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ["DLESM_BOARD_DIR"] = d
os.environ["DLESM_BOARD_TIMEOUT_S"] = "60"
from dl_esm_inf_amd import _cabi  # noqa: E402

L = _cabi.lib()
sid = C.create_string_buffer(session.encode(), _cabi.UNIQUE_ID_BYTES)
_cabi.check(L.dlesm_board_open(sid, n, rank))
assert L.dlesm_board_is_open() == 1
import time  # noqa: E402
for op, size in enumerate([8, 1, 1024, 0, 100000, 8, 8, 8]):
    if (op + rank) % 3 == 0:
        time.sleep(0.02 * rank)              # ranks out of step with each other
    mine = bytes((rank * 31 + op * 7 + k) % 251 for k in range(size))
    buf = C.create_string_buffer(mine, max(size, 1))
    every = C.create_string_buffer(max(size * n, 1))
    root_only = op == 4 and rank != 0         # a gather: the others only contribute
    _cabi.check(L.dlesm_board_allgather(buf, size, None if root_only else every))
    if not root_only:
        for r in range(n):
            want = bytes((r * 31 + op * 7 + k) % 251 for k in range(size))
            assert every.raw[r * size:(r + 1) * size] == want, (op, r)
# a sum in rank order, as dlesm_global_sum_f64 does it in mailbox mode
v = C.c_double(0.1 * (rank + 1))
allv = (C.c_double * n)()
_cabi.check(L.dlesm_board_allgather(C.byref(v), 8, allv))
s = 0.0
for r in range(n):
    s += allv[r]
want = 0.0
for r in range(n):
    want += 0.1 * (r + 1)
assert s == want
_cabi.check(L.dlesm_board_close())
assert L.dlesm_board_is_open() == 0
print(f"board rank {rank} ok")
