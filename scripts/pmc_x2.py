"""Only the launches whose fabric traffic is wanted (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; or under
--kernel-trace --stats): the two-steps-per-launch forms of the shallow-water update at 8192^2 -- NE offset plain / filtered, SW-offset
doubly periodic plain / filtered -- each beside its one-launch single step, six launches apiece with the time loop's rotation.
    python3 scripts/pmc_x2.py [tile]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"; D.parallel_init(0, 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew", "unew2", "vnew2", "pnew2"]
prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
P = D.psy
for offset, bc in ((D.GO_OFFSET_NE, (1, 1, 2)), (D.GO_OFFSET_SW, (D.GO_BC_PERIODIC, D.GO_BC_PERIODIC, D.GO_BC_NONE))):
    ne = offset == D.GO_OFFSET_NE
    g = D.grid_type(D.GO_ARAKAWA_C, bc, offset); g.decompose(N, N); D.grid_init(g, 1.0e5, 1.0e5)
    F = {}
    for k, nm in enumerate(names):
        F[nm] = D.r2d_field(g, pts[nm[0]])
        D.psy.hash_init(F[nm], 300 + k % 6)             # (the new levels start as copies of level n / n-1: same ring, valid halos)
        F[nm].data.mul_(0.01); F[nm].data.add_(1.0 if nm[0] == "p" else -0.005)
        if not ne:
            D.psy.apply_periodic_halos(F[nm])
    lv = [[F[n] for n in names[k:k + 3]] for k in (0, 3, 6, 9)]
    for filtered in (False, True):
        c, o, n1, n2 = lv
        for _ in range(6):                              # single steps
            if ne and not filtered: P.invoke_shallow_step(prm, *c, *o, *n1); o, c, n1 = c, n1, o
            elif ne: P.invoke_shallow_step_smooth(prm, 0.001, *c, *o, *n1); c, n1 = n1, c
            elif not filtered: P.invoke_shallow_step_sw_periodic(prm, *c, *o, *n1); o, c, n1 = c, n1, o
            else: P.invoke_shallow_step_sw_smooth_periodic(prm, 0.001, *c, *o, *n1); c, n1 = n1, c
        c, o, n1, n2 = lv
        for _ in range(6):                              # two steps per launch
            if ne and not filtered: P.invoke_shallow_step_x2(prm, *c, *o, *n1, *n2); c, o, n1, n2 = n2, n1, o, c
            elif ne: P.invoke_shallow_step_smooth_x2(prm, 0.001, *c, *o, *n1, *n2); c, o, n1, n2 = n1, n2, c, o
            elif not filtered: P.invoke_shallow_step_sw_x2_periodic(prm, *c, *o, *n1, *n2); c, o, n1, n2 = n2, n1, o, c
            else: P.invoke_shallow_step_sw_smooth_x2_periodic(prm, 0.001, *c, *o, *n1, *n2); c, o, n1, n2 = n1, n2, c, o
        torch.cuda.synchronize()
    del F, lv, c, o, n1, n2
    torch.cuda.empty_cache()
print("done")
