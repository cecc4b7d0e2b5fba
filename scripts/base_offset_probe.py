"""Does the RELATIVE placement of the nine shallow-water arrays in HBM matter?  (GPU box.)

The nine arrays of the fused step are nine concurrent streams; with every array the same size and
allocated back to back, stream k of a tile hits address base + k*size + off, and if `size` is a
multiple of the channel-interleave period all nine land on the same channel at the same moment.
This probe carves the nine arrays out of ONE buffer with a per-array stagger (array k starts
k*stagger bytes later than back-to-back placement would put it) and times the step for each
stagger.  Product calls only (the C ABI); no oracle.

    python scripts/base_offset_probe.py [--tile 8192] [--sw 0|1] [--what step|jacobi]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=8192)
ap.add_argument("--sw", type=int, default=0)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--what", default="step")
ap.add_argument("--staggers", default="")
ap.add_argument("--align2m", action="store_true", help="array-to-array distance = the size rounded up to 2 MiB (what an "
                "allocator that hands out 2 MiB-aligned blocks gives) + the stagger")
args = ap.parse_args()

import torch  # noqa: E402

import dl_esm_inf_amd as D  # noqa: E402

L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
N = args.tile
bc = (D.GO_BC_PERIODIC, D.GO_BC_PERIODIC, D.GO_BC_NONE) if args.sw else (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE)
g = D.grid_type(D.GO_ARAKAWA_C, bc, D.GO_OFFSET_SW if args.sw else D.GO_OFFSET_NE)
g.decompose(N, N)
D.grid_init(g, 1.0e5, 1.0e5)
ld, ny = g.nx, g.ny
t = D.r2d_field(g, D.GO_T_POINTS)
it = t.internal
box = (it.xstart, it.xstop, it.ystart, it.ystop)
del t
size = ld * ny * 8
SIZE_BYTES = size
if args.align2m:
    size = (size + (2 << 20) - 1) & ~((2 << 20) - 1)
prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
narr = 9 if args.what in ("step", "smooth") else 2
staggers = [int(x) for x in args.staggers.split(",")] if args.staggers else \
    [0, 256, 512, 1024, 2048, 4096, 4096 + 256, 8192, 8192 + 512, 16384, 32768, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256]
maxst = max(staggers)
buf = torch.empty((narr * (size + maxst) + 4096) // 8, dtype=torch.float64, device="cuda")
base = (buf.data_ptr() + 4095) & ~4095
print(f"tile {N} ld {ld} ny {ny} array {size} B = {size / 4096:.2f} x 4 KiB, base % 2 MiB = {base % (2 << 20)}; "
      f"size % 4096 = {size % 4096}, size % 65536 = {size % 65536}, size % 1 MiB = {size % (1 << 20)}", flush=True)
with torch.cuda.stream(s):
    buf.uniform_(0.5, 1.5)
s.synchronize()


def run(stagger):
    ptrs = [C.c_void_p(base + k * (size + stagger)) for k in range(narr)]
    if args.what == "smooth":      # the whole filtered step: cur <-> new alternate, old stays (and is updated in place)
        def once(rot):
            p = ptrs if rot % 2 == 0 else ptrs[6:] + ptrs[3:6] + ptrs[:3]
            rc = L.dlesm_shallow_step_smooth_f64(C.byref(prm), C.c_double(0.001), ld, ny, *box, *p, sp)
            assert rc == 0, D._cabi.last_error()
    elif args.what == "step":
        fn = L.dlesm_shallow_step_sw_f64 if args.sw else L.dlesm_shallow_step_f64

        def once(rot):
            p = ptrs[3 * rot:] + ptrs[:3 * rot]     # leapfrog rotation of the three levels
            rc = fn(C.byref(prm), ld, ny, *box, *p, sp)
            assert rc == 0, D._cabi.last_error()
    else:
        def once(rot):
            a, b = (ptrs[0], ptrs[1]) if rot % 2 == 0 else (ptrs[1], ptrs[0])
            rc = L.dlesm_stencil5_f64(a, b, ld, ny, *box, sp)
            assert rc == 0, D._cabi.last_error()
    best = 1e9
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            for i in range(args.reps):
                once(i % 3 if args.what == "step" else i)      # (smooth: i % 2 inside)
            e1.record(s)
        s.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1) / args.reps)
    return best


bpc = {"step": 72, "smooth": 96}.get(args.what, 16)
for st in staggers + staggers[:3]:
    ms = run(st)
    if args.what == "smooth":      # the two rotations separately: is one role assignment slower?
        save = args.reps
        both = []
        for par in (0, 1):
            ptrs_ = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pp = [C.c_void_p(base + k * (size + st)) for k in range(narr)]
            q = pp if par == 0 else pp[6:] + pp[3:6] + pp[:3]
            with torch.cuda.stream(s):
                e0.record(s)
                for _ in range(10):
                    L.dlesm_shallow_step_smooth_f64(C.byref(prm), C.c_double(0.001), ld, ny, *box, *q, sp)
                e1.record(s)
            s.synchronize()
            both.append(e0.elapsed_time(e1) / 10)
        print(f"      rotation 0: {both[0]:.4f} ms   rotation 1: {both[1]:.4f} ms", flush=True)
    print(f"stagger {st:8d} B  (array-to-array distance % 64 KiB = {(size + st) % 65536:6d}, % 4 KiB = {(size + st) % 4096:4d})  "
          f"{ms:.4f} ms  {bpc * N * N / ms / 1e6:7.1f} GB/s  frac {bpc * N * N / ms / 1e6 / 8000:.4f}", flush=True)
