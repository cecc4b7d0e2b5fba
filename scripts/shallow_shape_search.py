#!/usr/bin/env python3
"""Exhaustive (waves per workgroup, tiles per row) search for the fused shallow-water step at several sizes, beside the shape the
library's rule picks and the one its planning call picks: the data sw_rule_shape (dlesm_shallow.hip) is fitted to.
    python scripts/shallow_shape_search.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dl_esm_inf_amd as D
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
OUT_LANES = 56
sizes = [int(a) for a in sys.argv[1:]] or [2048, 3000, 4096, 6144, 8192, 10000, 12288]
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
for N in sizes:
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(N, N); D.grid_init(g, 1.0, 1.0)
    F = {}
    for k, n in enumerate(names):
        F[n] = D.r2d_field(g, pts[n[0]]); D.psy.hash_init(F[n], 5 + k)
        F[n].data.add_(1.0 if n[0] == "p" else -0.5)
    torch.cuda.synchronize()

    def run(**tune):
        base = dict(j5_autoshape=1, j5_tpb=0, j5_pad_tiles=0, j5_use_tuned=0)
        base.update(tune)
        for k, v in base.items():
            L.dlesm_set_tuning(k.encode(), v)
        cur, old, new = [F["u"], F["v"], F["p"]], [F["uold"], F["vold"], F["pold"]], [F["unew"], F["vnew"], F["pnew"]]
        ts = []
        reps = max(6, min(40, int(20 * (8192 / N) ** 2)))
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                D.psy.invoke_shallow_step(prm, *cur, *old, *new)
                old, cur, new = cur, new, old
            e1.record(); torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / reps)
        return min(ts)

    base = ((N + 1) // 2 - (1 // 2 & ~7) + OUT_LANES) // OUT_LANES
    rule = run()
    res = []
    for tpb in (2, 4, 8):
        for pad in range(0, 8 * tpb + 2):
            res.append((run(j5_autoshape=0, j5_tpb=tpb, j5_pad_tiles=pad), tpb, base + pad))
    res.sort()
    D.psy.autotune_shallow(prm, *[F[n] for n in names])
    planned = run(j5_use_tuned=1)
    print(f"N {N}: tiles per row unpadded {base}; rule {rule:.4f} ms; planned {planned:.4f} ms; best of the search:", flush=True)
    for ms, tpb, t in res[:8]:
        print(f"    {ms:.4f} ms  waves/group {tpb}  tiles/row {t:4d}  = {t / tpb:7.2f} groups/row  (mod 8: {(t / tpb) % 8:5.2f})", flush=True)
    worst = res[-1]
    print(f"    worst {worst[0]:.4f} ms (waves {worst[1]}, tiles {worst[2]})", flush=True)
    for k, v in dict(j5_autoshape=1, j5_tpb=0, j5_pad_tiles=0, j5_use_tuned=1).items():
        L.dlesm_set_tuning(k.encode(), v)
    del F
    torch.cuda.empty_cache()
