"""Only the launches whose fabric traffic is wanted (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): the whole
filtered leapfrog step (one launch, 96 B/cell) and the SW-offset periodic step (one launch, 72 B/cell) at 8192^2.
    python3 scripts/pmc_filtered.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
N = 8192
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
for offset, bc in ((D.GO_OFFSET_NE, (1, 1, 2)), (D.GO_OFFSET_SW, (D.GO_BC_PERIODIC, D.GO_BC_PERIODIC, D.GO_BC_NONE))):
    g = D.grid_type(D.GO_ARAKAWA_C, bc, offset); g.decompose(N, N); D.grid_init(g, 1.0e5, 1.0e5)
    F = {}
    for k, nm in enumerate(names):
        F[nm] = D.r2d_field(g, pts[nm[0]])
        D.psy.hash_init(F[nm], 300 + k)
        F[nm].data.mul_(0.01); F[nm].data.add_(1.0 if nm[0] == "p" else -0.005)
    cur, old, new = [F[n] for n in names[:3]], [F[n] for n in names[3:6]], [F[n] for n in names[6:]]
    for _ in range(6):
        if offset == D.GO_OFFSET_NE:
            D.psy.invoke_shallow_step_smooth(prm, 0.001, *cur, *old, *new)
        else:
            D.psy.invoke_shallow_step_sw_periodic(prm, *cur, *old, *new)
        cur, new = new, cur
    torch.cuda.synchronize()
    del F, cur, old, new
    torch.cuda.empty_cache()
print("done")
