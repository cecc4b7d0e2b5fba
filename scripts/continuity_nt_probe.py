#!/usr/bin/env python3
"""Cache policies of the continuity kernel (cont_nt 0..3), warm.   python scripts/continuity_nt_probe.py [tile]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, dl_esm_inf_amd as D
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
s = torch.cuda.Stream()
CF = [D.r2d_field(g, p) for p in (D.GO_T_POINTS, D.GO_T_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS)]
for k, f in enumerate(CF[1:]): D.psy.hash_init(f, 40 + k, stream=s)
g.area_t_device
def run(nt, tpbs=0):
    L.dlesm_set_tuning(b"cont_nt", nt)
    with torch.cuda.stream(s):
        for _ in range(10): D.psy.invoke_continuity(*CF, 0.5, stream=s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(60): D.psy.invoke_continuity(*CF, 0.5, stream=s)
        e1.record(s)
    s.synchronize()
    ms = e0.elapsed_time(e1) / 60
    return ms, 72 * tile * tile / ms / 1e6 / 80
for rep in range(2):
    for nt in (0, 1, 2, 3):
        ms, pc = run(nt); print(f"tile {tile} nt {nt}: {ms:.4f} ms {pc:.1f}%", flush=True)
