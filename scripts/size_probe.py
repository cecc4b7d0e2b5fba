#!/usr/bin/env python3
"""Jacobi-5 full-box time for several tile sizes / alignments / rows-per-tile with the automatic
block-shape rule (and forced alternatives), one process per size to keep memory bounded."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dl_esm_inf_amd as D
tile, align = int(sys.argv[1]), sys.argv[2]
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = align
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
box = a.internal.box()
s = torch.cuda.Stream(); sp = C.c_void_p(s.cuda_stream)
D.psy.hash_init(a, 1, stream=s); D.copy_field(a, b, stream=s)
with torch.cuda.stream(s):
    for R in (2, 3, 4):
        for tpb in (0, 2, 4, 8):
            L.dlesm_set_tuning(b"j5_tile_rows", R); L.dlesm_set_tuning(b"j5_tpb", tpb)
            x, y = a, b
            ts = []
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(20):
                    D._cabi.check(L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp))
                    x, y = y, x
                e1.record(s); s.synchronize()
                if rnd: ts.append(e0.elapsed_time(e1) / 20)
            t = min(ts)
            print(f"N {tile} A {align} ld {g.nx} R {R} tpb {tpb or 'auto'}: {t:.4f} ms  {16.0*tile*tile/t/1e9:6.0f} GB/s {16.0*tile*tile/t/1e9/80:.1f}%", flush=True)
