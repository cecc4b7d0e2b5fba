"""The whole filtered leapfrog step (one launch, 96 B/cell): its two rotations timed separately, per cache policy, over several fresh
sets of nine arrays -- is one role assignment systematically slower?   python scripts/smooth_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
N = 8192
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(N, N); D.grid_init(g, 1.0e5, 1.0e5)
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
s = torch.cuda.Stream()
def timed(fn, n=10):
    with torch.cuda.stream(s):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n): fn()
        e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) / n
keep = []
for rnd in range(4):
    F = {}
    with torch.cuda.stream(s):
        for k, nm in enumerate(names):
            F[nm] = D.r2d_field(g, pts[nm[0]])
            D.psy.hash_init(F[nm], 300 + k, stream=s)
            F[nm].data.mul_(0.01); F[nm].data.add_(1.0 if nm[0] == "p" else -0.005)
    s.synchronize()
    cur, old, new = [F[n] for n in names[:3]], [F[n] for n in names[3:6]], [F[n] for n in names[6:]]
    line = []
    for nt, ntl in ((2, 1), (2, 0), (0, 0), (3, 1)):
        L.dlesm_set_tuning(b"sw_nt", nt); L.dlesm_set_tuning(b"sw_smooth_ntl", ntl)
        a = timed(lambda: D.psy.invoke_shallow_step_smooth(prm, 0.001, *cur, *old, *new, stream=s))
        b = timed(lambda: D.psy.invoke_shallow_step_smooth(prm, 0.001, *new, *old, *cur, stream=s))
        line.append(f"nt={nt},ntl={ntl}: {a:.4f}/{b:.4f}")
    L.dlesm_set_tuning(b"sw_nt", 2); L.dlesm_set_tuning(b"sw_smooth_ntl", 1)
    print(f"set {rnd} (ms, rotation A / rotation B)  " + "   ".join(line), flush=True)
    if rnd % 2 == 0:
        keep.append(F)       # keep some sets alive so that the next ones land elsewhere
