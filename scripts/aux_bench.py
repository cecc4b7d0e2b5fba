#!/usr/bin/env python3
"""Rates of the other kernels around the hot path (one MI355X): masked Jacobi (20 B/cell), continuity (72 B/cell), checksum
(8 B/cell read), whole-field copy (16 B/cell), fill (8 B/cell written), hash init (8 B/cell written),
gather pack + unpack (16 B/cell each), periodic halo copies.   python scripts/aux_bench.py [tile]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dl_esm_inf_amd as D  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
it = a.internal
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
cells = tile * tile
D.psy.hash_init(a, 1, stream=s)
D.copy_field(a, b, stream=s)
slot = cells
send = torch.zeros(slot, dtype=torch.float64, device="cuda")
glob = torch.zeros(cells, dtype=torch.float64, device="cuda")
pd = g.decomp
val = C.c_double()


def timed(name, fn, bytes_per_cell, n=30):
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n):
            fn()
        e1.record(s)
    s.synchronize()
    ms = e0.elapsed_time(e1) / n
    gbs = bytes_per_cell * cells / ms / 1e6
    print(f"{name:46s} {ms:8.4f} ms  {gbs:7.0f} GB/s  ({gbs / 80:.1f}% of 8 TB/s; {bytes_per_cell} B/cell)", flush=True)


timed("jacobi5 (reference point; launch shape by rule)", lambda: D.psy.invoke_jacobi5(b, a, stream=s), 16)
if "--no-plan" not in sys.argv:
    # the optional planning call of an application (plan_jacobi5): the shape it measures for this geometry also serves the
    # 3x3, masked and continuity sweeps below (shape_for_tile_sweep)
    with torch.cuda.stream(s):
        D.psy.autotune_jacobi5(b, a, stream=s)
        D.copy_field(a, b, stream=s)
    s.synchronize()
    timed("jacobi5, planned shape", lambda: D.psy.invoke_jacobi5(b, a, stream=s), 16)
timed("stencil9 (general 3x3 weights)", lambda: D.psy.invoke_stencil9(b, a, [0.0625, 0.125, 0.0625, 0.125, 0.25, 0.125,
                                                                                  0.0625, 0.125, 0.0625], stream=s), 16)
timed("jacobi5 masked, all-wet mask", lambda: D.psy.invoke_jacobi5_masked(b, a, stream=s), 20)
CF = [D.r2d_field(g, p) for p in (D.GO_T_POINTS, D.GO_T_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS,
                                  D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS)]
for k, f in enumerate(CF[1:]):
    D.psy.hash_init(f, 40 + k, stream=s)
g.area_t_device
timed("continuity (T, U, V fields + grid%area_t)", lambda: D.psy.invoke_continuity(*CF, 0.5, stream=s), 72)
del CF
timed("copy_field (whole field)", lambda: D.copy_field(a, b, stream=s), 16)
timed("set_field (fill)", lambda: D.set_field(b, 1.0, stream=s), 8)
with torch.cuda.stream(s):
    timed("  write-only reference: torch fill_ of the same array", lambda: b.data.fill_(1.0), 8 * g.nx * g.ny / cells)
    timed("  write-only reference: torch zero_ (memset)", lambda: b.data.zero_(), 8 * g.nx * g.ny / cells)
L.dlesm_set_tuning(b"j5_nt_stores", 0)
timed("set_field, default stores", lambda: D.set_field(b, 1.0, stream=s), 8)
L.dlesm_set_tuning(b"j5_nt_stores", -1)
timed("hash_init", lambda: D.psy.hash_init(b, 7, stream=s), 8)
timed("field_checksum (device part + 8 B to host)", lambda: L.dlesm_checksum_f64(
    a.device_ptr, g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop, C.byref(val), sp), 8, n=10)
res_dev = torch.zeros(1, dtype=torch.float64, device="cuda")
timed("field_checksum, no host synchronisation (async entry)", lambda: L.dlesm_checksum_async_f64(
    a.device_ptr, g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop, C.c_void_p(res_dev.data_ptr()), sp), 8)
timed("gather: pack_inner", lambda: L.dlesm_pack_inner_f64(
    a.device_ptr, g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop, C.c_void_p(send.data_ptr()), slot, sp), 16)
timed("  pack of a box that starts on a 16-byte boundary", lambda: L.dlesm_pack_inner_f64(
    a.device_ptr, g.nx, g.ny, 1, tile, it.ystart, it.ystop, C.c_void_p(send.data_ptr()), slot, sp), 16)
timed("gather: unpack_gathered (1 rank)", lambda: L.dlesm_unpack_gathered_f64(
    C.c_void_p(send.data_ptr()), slot, C.byref(pd._info), pd.subdomains, 1, C.c_void_p(glob.data_ptr()), sp), 16, n=10)
# the pack linear in the dense destination (util_gather_linear = 2), then both copies as row segments (0): what the
# source-linear pack and the contiguous unpack replaced
L.dlesm_set_tuning(b"util_gather_linear", 2)
timed("gather: pack_inner, linear in the destination", lambda: L.dlesm_pack_inner_f64(
    a.device_ptr, g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop, C.c_void_p(send.data_ptr()), slot, sp), 16)
L.dlesm_set_tuning(b"util_gather_linear", 0)
timed("gather: pack_inner, row segments", lambda: L.dlesm_pack_inner_f64(
    a.device_ptr, g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop, C.c_void_p(send.data_ptr()), slot, sp), 16)
timed("gather: unpack_gathered (1 rank), row segments", lambda: L.dlesm_unpack_gathered_f64(
    C.c_void_p(send.data_ptr()), slot, C.byref(pd._info), pd.subdomains, 1, C.c_void_p(glob.data_ptr()), sp), 16, n=10)
L.dlesm_set_tuning(b"util_gather_linear", 1)
