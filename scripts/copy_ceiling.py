#!/usr/bin/env python3
"""Whole-array linear copy against the planned Jacobi sweep, warm (100 + 300 launches), with and without non-temporal stores.
   python scripts/copy_ceiling.py [tile]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, dl_esm_inf_amd as D
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
s = torch.cuda.Stream()
D.psy.hash_init(a, 1, stream=s)
def run(fn, nbytes, n=300, w=100):
    with torch.cuda.stream(s):
        for _ in range(w): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n): fn()
        e1.record(s)
    s.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms, nbytes / ms / 1e6
D.psy.autotune_jacobi5(b, a, stream=s)
for rep in range(2):
    for nt in (0, 1):
        L.dlesm_set_tuning(b"j5_nt_stores", nt)
        ms, gbs = run(lambda: D.copy_field(a, b, stream=s), 16 * g.nx * g.ny)
        print(f"tile {tile} nt_stores {nt} copy_field whole array ({g.nx}x{g.ny}): {ms:.4f} ms {gbs:.0f} GB/s {gbs/80:.1f}%", flush=True)
        x, y = [a], [b]
        def step():
            D.psy.invoke_jacobi5(y[0], x[0], stream=s); x[0], y[0] = y[0], x[0]
        ms, gbs = run(step, 16 * tile * tile)
        print(f"tile {tile} nt_stores {nt} jacobi5: {ms:.4f} ms {gbs:.0f} GB/s {gbs/80:.1f}%", flush=True)
