#!/usr/bin/env python3
"""experiment: shallow-water tile kernel vs block shape / padding; direct kernel as reference"""
import ctypes as C, os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dl_esm_inf_amd as D
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
s = torch.cuda.Stream()
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
F = {}
with torch.cuda.stream(s):
    for k, n in enumerate(names):
        F[n] = D.r2d_field(g, pts[n[0]]); D.psy.hash_init(F[n], 5 + k, stream=s)
        F[n].data.add_(1.0 if n[0] == "p" else -0.5)
prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
cells = tile * tile
def run(label, **tune):
    base = dict(sw_kernel=0, sw_tile_rows=2, sw_dpp=1, j5_autoshape=1, j5_tpb=0, j5_pad_tiles=0, j5_skew=1)
    base.update(tune)
    for k, v in base.items(): L.dlesm_set_tuning(k.encode(), v)
    cur, old, new = [F["u"], F["v"], F["p"]], [F["uold"], F["vold"], F["pold"]], [F["unew"], F["vnew"], F["pnew"]]
    ts = []
    with torch.cuda.stream(s):
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20):
                D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=s)
                old, cur, new = cur, new, old
            e1.record(s); s.synchronize()
            if rnd: ts.append(e0.elapsed_time(e1) / 20)
    ms = min(ts)
    print(f"{label:44s} {ms:.4f} ms  {72.0*cells/ms/1e6:6.0f} GB/s  {72.0*cells/ms/1e6/80:.1f}%", flush=True)
run("direct", sw_kernel=1)
for rep in range(2):
    for R in (2, 3, 1):
        run(f"tile auto R{R} dpp", sw_tile_rows=R, sw_dpp=1)
        run(f"tile auto R{R} bpermute", sw_tile_rows=R, sw_dpp=0)
if len(sys.argv) > 2:      # exhaustive (waves per group, tiles per row) search, as scripts/shape_search.py
    base = (tile // 2 + 62) // 62
    print(f"tiles per row without padding: {base}")
    for tpb in (4, 8):
        for pad in range(0, 8 * tpb + 4):
            run(f"plain tpb {tpb} pad {pad:2d} -> {base + pad:3d} tiles = {(base + pad) / tpb:6.2f} groups/row", j5_autoshape=0,
                j5_tpb=tpb, j5_pad_tiles=pad)
