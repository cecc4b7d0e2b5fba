"""continuity kernel (8 arrays read + 1 written, 72 B/cell) against the 8+1 stream-copy ceiling of the same box, per store/load policy.
    python scripts/continuity_probe.py [tile]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
s = torch.cuda.Stream(); cells = tile * tile
NDUMMY = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # arrays of the field's size allocated BEFORE the nine of the kernel
dummies = [torch.zeros((g.ny, g.nx), dtype=torch.float64, device="cuda") for _ in range(NDUMMY)]
CF = [D.r2d_field(g, p) for p in (D.GO_T_POINTS, D.GO_T_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS, D.GO_U_POINTS, D.GO_V_POINTS)]
for k, f in enumerate(CF[1:]): D.psy.hash_init(f, 40 + k, stream=s)
g.area_t_device
print("base addresses mod 2 MiB / 1 GiB:", [(f.data.data_ptr() % (2 << 20), (f.data.data_ptr() >> 30)) for f in CF], g.area_t_device.data_ptr() % (2 << 20), flush=True)
def timed(fn, n=20):
    best = 1e9
    for r in range(3):
        with torch.cuda.stream(s):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(n): fn()
            e1.record(s)
        s.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return best
sp = C.c_void_p(s.cuda_stream)
n = (g.nx * g.ny) & ~1
srcs = (C.c_void_p * 8)(*[f.device_ptr for f in CF[1:]] + [g.area_t_device.data_ptr()])
dsts = (C.c_void_p * 1)(CF[0].device_ptr)
for nt in (0, 2):
    ms = timed(lambda: D._cabi.check_lab(D._cabi.lab().dlesm_lab_stream_copy_f64(8, 1, srcs, dsts, n, nt, sp)))
    print(f"{tile}^2 stream copy 8 read + 1 written, nt={nt}: {ms:.4f} ms  {72 * n / ms / 1e6 / 80:.1f} % of 8 TB/s", flush=True)
for nt in (0, 1, 2, 3):
    L.dlesm_set_tuning(b"cont_nt", nt)
    ms = timed(lambda: D.psy.invoke_continuity(*CF, 0.5, stream=s))
    print(f"{tile}^2 continuity cont_nt={nt}: {ms:.4f} ms  {72 * cells / ms / 1e6 / 80:.1f} % of 8 TB/s", flush=True)
