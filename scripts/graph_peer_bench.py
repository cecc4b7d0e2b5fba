#!/usr/bin/env python3
"""Host issue cost of a distributed time loop over the mailboxes, step by step from Python against ONE hipGraph of two
steps replayed (loop-back on one GPU; the steps' sequence numbers live on the device).
    python scripts/graph_peer_bench.py [tile=1024] [steps=400]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
import dl_esm_inf_amd as D  # noqa: E402
from dm_overhead import loopback_tables  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
x, y = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
it = x.internal
t = loopback_tables(D, it)
plan = C.c_void_p()
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
D._cabi.check(L.dlesm_halo_plan_peer_connect_rccl(plan, 1))
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
box = it.box()


def pair(fn):
    D._cabi.check(fn(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp))
    D._cabi.check(fn(plan, y.device_ptr, x.device_ptr, g.nx, g.ny, *box, sp))


out = {"tile": tile, "steps": steps, "what": "us per step, wall clock of the issuing thread incl. the final synchronise"}
with torch.cuda.stream(s):
    D.psy.hash_init(x, 7, stream=s)
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, sp))
    D.copy_field(x, y, stream=s)
    pair(L.dlesm_jacobi5_step_dm_pipelined)
    D._cabi.check(L.dlesm_halo_plan_join(plan, sp))
    pair(L.dlesm_jacobi5_step_dm)                                          # warm both forms
s.synchronize()
for name, fn in (("time_loop", L.dlesm_jacobi5_step_dm_pipelined), ("joined", L.dlesm_jacobi5_step_dm)):
    # eager
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // 2):
        pair(fn)
    D._cabi.check(L.dlesm_halo_plan_join(plan, sp))
    s.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e6
    out[name] = {"python_step_by_step_us": round(eager, 2)}
    for pairs in (1, 10):                # steps per graph: 2, 20
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            for _ in range(pairs):
                pair(fn)
            D._cabi.check(L.dlesm_halo_plan_join(plan, sp))
        with torch.cuda.stream(s):
            graph.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps // (2 * pairs)):
                graph.replay()
            s.synchronize()
        replay = (time.perf_counter() - t0) / (steps // (2 * pairs) * 2 * pairs) * 1e6
        out[name][f"graph_of_{2 * pairs}_steps_us"] = round(replay, 2)
        del graph
# the plain sweep for scale (no exchange), eager
s.synchronize()
t0 = time.perf_counter()
for _ in range(steps // 2):
    D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp))
    D._cabi.check(L.dlesm_stencil5_f64(y.device_ptr, x.device_ptr, g.nx, g.ny, *box, sp))
s.synchronize()
out["plain_sweep_python_us"] = round((time.perf_counter() - t0) / steps * 1e6, 2)
assert L.dlesm_wait_timed_out(0) == 0
D._cabi.check(L.dlesm_halo_plan_destroy(plan))
print(json.dumps(out, indent=1))
