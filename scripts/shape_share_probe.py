"""Do the other two-row wave-tile sweeps (3x3, masked, continuity) like the launch shape the Jacobi planner picks for the same geometry?
    python scripts/shape_share_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
for tile in (8192, 16384):
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    s = torch.cuda.Stream(); cells = tile * tile
    D.psy.hash_init(a, 1, stream=s)
    CF = [D.r2d_field(g, p) for p in (D.GO_T_POINTS, D.GO_T_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS)]
    for k, f in enumerate(CF[1:]): D.psy.hash_init(f, 40 + k, stream=s)
    g.area_t_device
    def timed(fn, bpc, n=20):
        best = 1e9
        for r in range(3):
            with torch.cuda.stream(s):
                for _ in range(3): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(n): fn()
                e1.record(s)
            s.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
        return bpc * cells / best / 1e6 / 80
    coef = [0.0625, 0.125, 0.0625, 0.125, 0.25, 0.125, 0.0625, 0.125, 0.0625]
    with torch.cuda.stream(s):
        D.psy.autotune_jacobi5(b, a, stream=s)
    s.synchronize()
    shape = D.psy.planned_shape_jacobi5(b)           # waves per group, tiles per row, rows, nt
    nxw0 = (tile // 2 + 64) // 64
    kernels = (("jacobi5", lambda: D.psy.invoke_jacobi5(b, a, stream=s), 16), ("stencil9", lambda: D.psy.invoke_stencil9(b, a, coef, stream=s), 16),
               ("masked", lambda: D.psy.invoke_jacobi5_masked(b, a, stream=s), 20), ("continuity", lambda: D.psy.invoke_continuity(*CF, 0.5, stream=s), 72))
    for name, fn, bpc in kernels:
        L.dlesm_set_tuning(b"j5_autoshape", 1); L.dlesm_set_tuning(b"j5_tpb", 0); L.dlesm_set_tuning(b"j5_pad_tiles", 0)
        rule = timed(fn, bpc)
        L.dlesm_set_tuning(b"j5_autoshape", 0); L.dlesm_set_tuning(b"j5_tpb", shape[0]); L.dlesm_set_tuning(b"j5_pad_tiles", shape[1] - nxw0)
        forced = timed(fn, bpc)
        L.dlesm_set_tuning(b"j5_autoshape", 1); L.dlesm_set_tuning(b"j5_tpb", 0); L.dlesm_set_tuning(b"j5_pad_tiles", 0)
        print(f"{tile}^2 {name:10s} rule {rule:5.1f} %   Jacobi's planned shape {list(shape)} forced {forced:5.1f} %", flush=True)
    del CF, a, b
    torch.cuda.empty_cache()
