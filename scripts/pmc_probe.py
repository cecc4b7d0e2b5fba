#!/usr/bin/env python3
"""Launch a fixed sequence of kernels with KNOWN byte counts next to the Jacobi kernel so that
a rocprofv3 --pmc pass can be calibrated (guide: calibrate FETCH_SIZE/WRITE_SIZE on a known
byte count in your own access pattern).  Prints the launch sequence; run under rocprofv3:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 scripts/pmc_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dl_esm_inf_amd as D  # noqa: E402

tile = int(os.environ.get("PROBE_TILE", "16384"))
seq = os.environ.get("PROBE_SEQ", "tcopy,pcopy,j8,j64,j265,j64v4").split(",")
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
D.psy.hash_init(a, 20261004)
torch.cuda.synchronize()
print(f"field bytes {g.nx * g.ny * 8}  interior bytes {tile * tile * 8}")
for tag in seq:
    for rep in range(3):
        if tag == "tcopy":
            b.data.copy_(a.data)
        elif tag == "pcopy":
            D.copy_field(a, b)
        elif tag.startswith("j"):
            rows, _, var = tag[1:].partition("v")
            L.dlesm_set_tuning(b"j5_rows", int(rows))
            L.dlesm_set_tuning(b"j5_variant", int(var or 0))
            D.psy.invoke_jacobi5(b, a)
        torch.cuda.synchronize()
    print("launched 3x", tag)
