#!/usr/bin/env python3
"""Is an IN-PLACE stream (an array both read and written, as time_smooth's field_old) slower than the same bytes with a separate
output?  Linear sweeps through dlesm_lab_stream_copy_f64, 3 arrays read + 1 written, 8192^2 field shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0)
n = 8256 * 8195
a, b, c, d = (torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(4))
s = torch.cuda.Stream(); sp = C.c_void_p(s.cuda_stream)
def timed(name, src, dst, nt, reps=30):
    sa = (C.c_void_p * len(src))(*[t.data_ptr() for t in src]); da = (C.c_void_p * 1)(dst.data_ptr())
    best = 1e9
    for rnd in range(4):
        with torch.cuda.stream(s):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                D._cabi.check_lab(D._cabi.lab().dlesm_lab_stream_copy_f64(len(src), 1, sa, da, n, nt, sp))
            e1.record(s)
        s.synchronize()
        if rnd: best = min(best, e0.elapsed_time(e1) / reps)
    gbs = 8.0 * (len(src) + 1) * n / best / 1e6
    print(f"{name:58s} {best:.4f} ms {gbs:7.0f} GB/s {gbs/80:.1f} %", flush=True)
for nt in (0, 2):
    timed(f"3r+1w, separate output, nt={nt}", [a, b, c], d, nt)
    timed(f"3r+1w, output = third read array (in place), nt={nt}", [a, b, c], c, nt)
    timed(f"3r+1w, output = first read array (in place), nt={nt}", [a, b, c], a, nt)
    timed(f"1r+1w separate, nt={nt}", [a], d, nt)
    timed(f"1r+1w in place, nt={nt}", [a], a, nt)
