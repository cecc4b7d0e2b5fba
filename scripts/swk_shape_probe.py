"""The seven GOcean kernels one by one (224 B/cell) at 8192^2: the rule's launch shape against forced (waves per group, tiles per row).
    python scripts/swk_shape_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
N = 8192
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(N, N); D.grid_init(g, 1.0e5, 1.0e5)
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
s = torch.cuda.Stream()
F = {}
with torch.cuda.stream(s):
    for k, nm in enumerate(names):
        F[nm] = D.r2d_field(g, pts[nm[0]]); D.psy.hash_init(F[nm], 300 + k, stream=s)
        F[nm].data.mul_(0.01); F[nm].data.add_(1.0 if nm[0] == "p" else -0.005)
    I = [D.r2d_field(g, t) for t in (D.GO_U_POINTS, D.GO_V_POINTS, D.GO_F_POINTS, D.GO_T_POINTS)]
s.synchronize()
prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
def seq():
    D.psy.invoke_shallow_kernel_sequence(40.0, *[F[n] for n in names[:6]], *I, *[F[n] for n in names[6:]], stream=s)
def timed(n=8):
    best = 1e9
    for r in range(3):
        with torch.cuda.stream(s):
            seq()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(n): seq()
            e1.record(s)
        s.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return 224 * N * N / best / 1e6 / 80
print(f"rule: {timed():.2f} %", flush=True)
with torch.cuda.stream(s):
    D.psy.autotune_shallow(prm, *[F[n] for n in names], stream=s)
s.synchronize()
print(f"rule, after the fused step's planning call: {timed():.2f} %", flush=True)
nxw0 = (N // 2 + 1 + 64) // 64
L.dlesm_set_tuning(b"j5_autoshape", 0)
for tpb in (4, 8):
    line = []
    for pad in (0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
        L.dlesm_set_tuning(b"j5_tpb", tpb); L.dlesm_set_tuning(b"j5_pad_tiles", pad)
        line.append(f"{nxw0 + pad}:{timed():.1f}")
    print(f"waves per group {tpb}, tiles per row: " + "  ".join(line), flush=True)
L.dlesm_set_tuning(b"j5_autoshape", 1); L.dlesm_set_tuning(b"j5_tpb", 0); L.dlesm_set_tuning(b"j5_pad_tiles", 0)
