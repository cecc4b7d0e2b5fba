#!/usr/bin/env python3
"""Where does the distributed step's overhead come from?  Times, on one GPU, (a) the full-box
stencil, (b) the interior-box stencil alone, (c) frame + interior on one stream, (d) the same
with an idle event round-trip through the side stream, (e) the full overlapped step."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
import dl_esm_inf_amd as D  # noqa: E402
from dm_overhead import loopback_tables  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = 40
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
it = a.internal
plan = C.c_void_p()
t = loopback_tables(D, it)
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
s = torch.cuda.Stream()
side = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
xs, xe, ys, ye = it.box()
D.psy.hash_init(a, 1, stream=s)
D.copy_field(a, b, stream=s)


def full(x, y):
    D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, xs, xe, ys, ye, sp))


def interior(x, y):
    D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, xs + 1, xe - 1, ys + 1, ye - 1, sp))


def shifted_y(x, y):
    D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, xs, xe, ys + 1, ye - 1, sp))


def shifted_x(x, y):
    D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, xs + 1, xe - 1, ys, ye, sp))


def exch_then_full(x, y):
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_EDGES_ONLY, C.c_void_p(side.cuda_stream)))
    full(x, y)


def dm(x, y):
    D._cabi.check(L.dlesm_jacobi5_step_dm(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, xs, xe, ys, ye, sp))


def dm_pipelined(x, y):
    D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, xs, xe, ys, ye, sp))


def exch_alone(x, y):
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_EDGES_ONLY, sp))


def rccl_beside_interior(x, y):
    # the exchange of x's halos on the side stream while the INTERIOR box streams: no frame work at all
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_EDGES_ONLY, C.c_void_p(side.cuda_stream)))
    interior(x, y)


def ns_only_beside_interior(x, y):     # two in-place row messages: RCCL alone, no pack / unpack kernels
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, 0xC | D._cabi.DIRS_NO_DIAGONALS, C.c_void_p(side.cuda_stream)))
    interior(x, y)


def ew_only_beside_interior(x, y):     # two column messages: pack + RCCL + unpack
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, 0x3 | D._cabi.DIRS_NO_DIAGONALS, C.c_void_p(side.cuda_stream)))
    interior(x, y)


def tiny_kernels_beside_interior(x, y):   # the side stream runs two small copy kernels only (no RCCL)
    with torch.cuda.stream(side):
        D.copy_field(x, src=D._cabi.Region(1, 8192, xs, xs, ys, ye), dest=D._cabi.Region(1, 8192, xs - 1, xs - 1, ys, ye), stream=side)
        D.copy_field(x, src=D._cabi.Region(1, 8192, xe, xe, ys, ye), dest=D._cabi.Region(1, 8192, xe + 1, xe + 1, ys, ye), stream=side)
    interior(x, y)


with torch.cuda.stream(s):
    for name, fn in (("full box", full), ("interior box", interior), ("y-shifted box", shifted_y),
                     ("x-shifted box", shifted_x), ("exchange on side stream || full box", exch_then_full),
                     ("exchange on side stream || interior box", rccl_beside_interior),
                     ("N/S messages only on side stream || interior box", ns_only_beside_interior),
                     ("E/W messages only on side stream || interior box", ew_only_beside_interior),
                     ("two column-copy kernels on side stream || interior", tiny_kernels_beside_interior),
                     ("exchange alone", exch_alone),
                     ("overlapped dm step", dm), ("pipelined dm step", dm_pipelined), ("full box again", full),
                     ("interior box again", interior)):
        x, y = a, b
        for _ in range(5):
            fn(x, y)
            x, y = y, x
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(steps):
            fn(x, y)
            x, y = y, x
        if fn is dm_pipelined:
            D._cabi.check(L.dlesm_halo_plan_join(plan, sp))
        e1.record(s)
        torch.cuda.synchronize()
        print(f"{name:40s} {e0.elapsed_time(e1) / steps:.4f} ms/step", flush=True)
