#!/usr/bin/env python3
"""linear fill (whole rows) against row-segment fill (a box) of the same 16384^2 field, five launches each: for rocprofv3 --pmc"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(16384, 16384); D.grid_init(g, 1.0, 1.0)
b = D.r2d_field(g, D.GO_T_POINTS); it = b.internal
for _ in range(5):
    D._cabi.check(L.dlesm_fill_f64(b.device_ptr, g.nx, g.ny, 1, g.nx, it.ystart, it.ystop, 1.0, None))
for _ in range(5):
    D._cabi.check(L.dlesm_fill_f64(b.device_ptr, g.nx, g.ny, 1, g.nx - 1, it.ystart, it.ystop, 1.0, None))
torch.cuda.synchronize()
