"""stencil9 / masked Jacobi / Jacobi (rule shape, no planning) at 8192^2 and 16384^2: % of 8 TB/s.   python scripts/aux_quick.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
for tile in (8192, 16384):
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    s = torch.cuda.Stream(); cells = tile * tile
    D.psy.hash_init(a, 1, stream=s)
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
    print(tile, "jacobi5 %.1f" % timed(lambda: D.psy.invoke_jacobi5(b, a, stream=s), 16),
          "stencil9 %.1f" % timed(lambda: D.psy.invoke_stencil9(b, a, coef, stream=s), 16),
          "masked %.1f" % timed(lambda: D.psy.invoke_jacobi5_masked(b, a, stream=s), 20), flush=True)
    del a, b
    torch.cuda.empty_cache()
