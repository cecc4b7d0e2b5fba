#!/usr/bin/env python3
"""experiment: exhaustive search over (waves per workgroup, idle padding tiles per row) for the
Jacobi-5 tile sweep at one size, to refit choose_block_shape():  scripts/shape_search.py N [A [T [ROWS]]]
(T > 1: the fused T-step kernel instead of the single-step one; ROWS: rows per wave tile of the single-step sweep, 2 or 3)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dl_esm_inf_amd as D  # noqa: E402

tile = int(sys.argv[1])
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = sys.argv[2] if len(sys.argv) > 2 else "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
box = a.internal.box()
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
D.psy.hash_init(a, 1, stream=s)
D.copy_field(a, b, stream=s)
FUSED = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 4:
    L.dlesm_set_tuning(b"j5_tile_rows", int(sys.argv[4]))
OL = 64 - 2 * ((FUSED + 1) // 2) if FUSED > 1 else 64
base = (box[1] // 2 - 0 + OL) // OL          # wave tiles per row, tile origin at chunk 0 (x0 = 1)


def run(**kw):
    for k, v in kw.items():
        L.dlesm_set_tuning(k.encode(), v)
    x, y = a, b
    ts = []
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            if FUSED > 1:
                D._cabi.check(L.dlesm_stencil5_multi_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, FUSED, *box, *box,
                                                         0, 0, 0, 0, sp))
            else:
                D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp))
            x, y = y, x
        e1.record(s)
        s.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 10)
    return min(ts)


with torch.cuda.stream(s):
    auto = run(j5_autoshape=1, j5_tpb=0, j5_pad_tiles=0)
    rows = []
    for tpb in ((4, 8) if FUSED > 1 else (2, 4, 8)):
        for pad in range(0, 8 * tpb + 2):
            rows.append((run(j5_autoshape=0, j5_tpb=tpb, j5_pad_tiles=pad), tpb, pad))
gb = 16.0 * tile * tile / 1e9
print(f"N {tile} fused {FUSED} ld {g.nx} tiles/row {base}: auto rule {auto:.4f} ms ({gb / auto / 8:.1f} %)")
for t, tpb, pad in sorted(rows)[:8]:
    print(f"   best: tpb {tpb} pad {pad:2d} -> {base + pad:4d} tiles = {(base + pad) / tpb:7.2f} groups/row  {t:.4f} ms ({gb / t / 8:.1f} %)")
for tpb in ((4, 8) if FUSED > 1 else (2, 4, 8)):
    line = " ".join(f"{t:.3f}" for t, tp, pad in rows if tp == tpb)
    print(f"   tpb {tpb} pad 0..: {line}")
