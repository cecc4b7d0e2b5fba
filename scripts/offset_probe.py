#!/usr/bin/env python3
"""experiment: does the distance (mod the DRAM interleave) between the `in` and `out` arrays of the
Jacobi sweep matter?  Both arrays are carved out of one allocation; `out` starts at
round_up(field bytes, 2 MiB) + delta for a list of deltas."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dl_esm_inf_amd as D  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
f = D.r2d_field(g, D.GO_T_POINTS)
box = f.internal.box()
nbytes = g.nx * g.ny * 8
MB2 = 2 << 20
span = (nbytes + MB2 - 1) // MB2 * MB2
buf = torch.rand((2 * span + (64 << 20)) // 8, dtype=torch.float64, device="cuda")
base = buf.data_ptr()
base_al = (base + MB2 - 1) // MB2 * MB2
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
deltas = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20,
          3 << 19, (1 << 20) + 4096, 131584, 65792]
res = {d: [] for d in deltas}
with torch.cuda.stream(s):
    for rnd in range(4):
        for d in deltas:
            pa, pb = C.c_void_p(base_al), C.c_void_p(base_al + span + d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                D._cabi.check(L.dlesm_stencil5_f64(pa, pb, g.nx, g.ny, *box, sp))
                D._cabi.check(L.dlesm_stencil5_f64(pb, pa, g.nx, g.ny, *box, sp))
            e1.record(s)
            s.synchronize()
            if rnd:
                res[d].append(e0.elapsed_time(e1) / 20)
print(f"tile {tile}, field {nbytes} B, out = in + {span} + delta")
for d in deltas:
    m = statistics.median(res[d])
    print(f"  delta {d:8d} B   {m:.4f} ms   {16.0 * tile * tile / m / 1e6:6.0f} GB/s")
