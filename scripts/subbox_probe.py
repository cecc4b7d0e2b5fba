"""set_field / hash_init restricted to boxes of a 16384^2 field (ld 16448): % of 8 TB/s per box and writer form
(util_rowlinear 1 = the linear sweep over whole rows that masks the columns outside the box; 0 = row segments).
    python scripts/subbox_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(16384, 16384); D.grid_init(g, 1.0, 1.0)
b = D.r2d_field(g, D.GO_T_POINTS); it = b.internal
def timed(fn, n=20):
    best = 1e9
    for r in range(3):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
boxes = {"whole rows": (1, g.nx), "internal 2..16385": (it.xstart, it.xstop), "1..16447": (1, g.nx - 1), "17..16400": (17, 16400),
         "1..12400 (3/4)": (1, 12400), "1..8192 (half)": (1, 8192)}
for name, (x0, x1) in boxes.items():
    cells = (x1 - x0 + 1) * it.ny
    line = []
    for rl in (1, 0):
        L.dlesm_set_tuning(b"util_rowlinear", rl)
        ms = timed(lambda: D._cabi.check(L.dlesm_fill_f64(b.device_ptr, g.nx, g.ny, x0, x1, it.ystart, it.ystop, 1.0, None)))
        mh = timed(lambda: D._cabi.check(L.dlesm_hash_init_f64(b.device_ptr, g.nx, g.ny, x0, x1, it.ystart, it.ystop, 7, 0, 0, None)))
        line.append(f"rowlinear={rl}: fill {8 * cells / ms / 1e6 / 80:5.1f} %  hash {8 * cells / mh / 1e6 / 80:5.1f} %")
    print(f"{name:22s} " + "   ".join(line), flush=True)
L.dlesm_set_tuning(b"util_rowlinear", 1)
