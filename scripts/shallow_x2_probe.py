#!/usr/bin/env python3
"""Time dlesm_shallow_step_x2_f64 (two leapfrog steps per launch, 96 B/cell per launch) beside the single fused step
(72 B/cell) at one size, same process, interleaved passes.

    python scripts/shallow_x2_probe.py [N] [--tune KEY=INT ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
    import torch
    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    torch.cuda.set_device(0)
    os.environ["DL_ESM_ALIGNMENT"] = "64"
    D.parallel_init(0, 1)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(n, n)
    D.grid_init(g, 1.0, 1.0)
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew", "unew2", "vnew2", "pnew2"]
    F = [D.r2d_field(g, D.GO_T_POINTS) for _ in names]
    for k, f in enumerate(F):
        D.psy.hash_init(f, 77 + k % 6)
        f.data.mul_(0.1)
        f.data.add_(1.0 if k % 3 == 2 else -0.05)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
    s = torch.cuda.Stream()
    D.psy.autotune_shallow(prm, *F[:9], stream=s)
    cells = n * n

    def timed(fn, reps):
        with torch.cuda.stream(s):
            for _ in range(4):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                fn()
            e1.record(s)
        s.synchronize()
        return e0.elapsed_time(e1) / reps

    single = lambda: D.psy.invoke_shallow_step(prm, *F[:9], stream=s)           # noqa: E731
    res = {"tile": n, "single_ms": [], "variants": {}}
    # (every variant but the first needs DLESM_HIP_LIB=dl_esm_inf_amd/lib/libdlesm_hip_lab.so; the product runs them all as the default)
    variants = [("R4 nt2 stack4 (the product's)", 4, 2, 4, 0), ("R4 nt2, tiles dealt row-major", 4, 2, 0, 0), ("R4 nt3 stack4 (nt loads of level n-1)", 4, 3, 4, 0),
                ("R4 nt6 stack4 (all loads first)", 4, 6, 4, 0), ("R2 nt2 stack4", 2, 2, 4, 0), ("R6 nt6 stack4", 6, 6, 4, 0), ("R4 nt2 stack8", 4, 2, 8, 0),
                ("R4 nt2 stack2", 4, 2, 2, 0), ("R4 nt0 stack4 (cached stores)", 4, 0, 4, 0), ("R4 nt2 stack4 pad1", 4, 2, 4, 1)]
    for rep in range(2):
        res["single_ms"].append(timed(single, 30))
        for name, rows, nt, stack, pad in variants:
            L.dlesm_set_tuning(b"sw_x2_rows", rows)
            L.dlesm_set_tuning(b"sw_x2_nt", nt)
            L.dlesm_set_tuning(b"sw_x2_stack", stack)
            L.dlesm_set_tuning(b"sw_x2_pad", pad)
            ms = timed(lambda: D.psy.invoke_shallow_step_x2(prm, *F, stream=s), 30)
            res["variants"].setdefault(name, []).append(ms)
    import ctypes as C
    lab = D._cabi.lab()
    src = (C.c_void_p * 6)(*[f.device_ptr_data() if hasattr(f, "device_ptr_data") else f.data.data_ptr() for f in F[:6]])
    dst = (C.c_void_p * 6)(*[f.data.data_ptr() for f in F[6:]])
    nd = (g.nx * g.ny) & ~1
    for nt_, label in ((0, "copy_6r6w_default_stores_ms"), (2, "copy_6r6w_nt_stores_ms")):
        res[label] = timed(lambda: D._cabi.check_lab(lab.dlesm_lab_stream_copy_f64(6, 6, src, dst, nd, nt_, C.c_void_p(s.cuda_stream))), 30)
    sm = min(res["single_ms"])
    res["single"] = {"ms_per_step": sm, "mcells_per_s": cells / sm / 1e3, "frac_of_peak_72B": 72 * cells / (sm * 1e-3) / 8e12}
    for name, v in res["variants"].items():
        ms = min(v)
        res["variants"][name] = {"ms_per_launch": ms, "ms_per_step": ms / 2, "mcells_per_s": 2 * cells / ms / 1e3,
                                 "frac_of_peak_96B_per_launch": 96 * cells / (ms * 1e-3) / 8e12, "speedup_per_step": 2 * sm / ms}
    best_copy = min(res["copy_6r6w_default_stores_ms"], res["copy_6r6w_nt_stores_ms"])
    res["copy_6r6w_frac_of_peak"] = 96 * nd / (best_copy * 1e-3) / 8e12
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
