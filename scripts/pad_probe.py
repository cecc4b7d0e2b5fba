#!/usr/bin/env python3
"""experiment: interior-box stencil time vs number of idle padding tiles per row"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dl_esm_inf_amd as D
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = sys.argv[2] if len(sys.argv) > 2 else "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
xs, xe, ys, ye = a.internal.box()
s = torch.cuda.Stream(); sp = C.c_void_p(s.cuda_stream)
D.psy.hash_init(a, 1, stream=s); D.copy_field(a, b, stream=s)
with torch.cuda.stream(s):
    for box_name, box in (("full", (xs, xe, ys, ye)), ("interior", (xs + 1, xe - 1, ys + 1, ye - 1))):
        for tpb, skew in ((2, 0), (4, 0), (4, 1), (8, 0), (8, 1), (16, 0), (16, 1), (0, 1)):
            pad = f"tpb{tpb} skew{skew}"
            L.dlesm_set_tuning(b"j5_tpb", tpb)
            L.dlesm_set_tuning(b"j5_skew", skew)
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
            print(f"ld {g.nx} {box_name:9s} pad {pad}: {min(ts):.4f} ms", flush=True)
