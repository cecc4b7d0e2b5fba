#!/usr/bin/env python3
"""Where the overhead of the distributed shallow-water step goes: time-loop form with the RCCL group, the unpack, the frame
columns, all frame cells switched off in turn (dm_skip_parts / sw_dm_diag diagnostics; results are then wrong), against the plain
step and the plain kernel on the interior box.   python scripts/shallow_dm_parts.py [tile]"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, dl_esm_inf_amd as D
from dm_overhead import loopback_tables
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; steps = 40
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
t = loopback_tables(D, F["p"].internal); plan = C.c_void_p()
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan))); g._halo_plan = plan
s = torch.cuda.Stream(); prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
def run(kind, skip, diag=0):
    L.dlesm_set_tuning(b"dm_skip_parts", skip)
    L.dlesm_set_tuning(b"sw_dm_diag", diag)
    cur, old, new = [F[n] for n in names[:3]], [F[n] for n in names[3:6]], [F[n] for n in names[6:]]
    with torch.cuda.stream(s):
        for k, n in enumerate(names):
            D.set_field(F[n], 1.0 if n[0] == "p" else 0.0, stream=s)
        for phase in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(steps if phase else 8):
                if kind == "plain": D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=s)
                elif kind == "interior":
                    it = cur[2].internal
                    D._cabi.check(L.dlesm_shallow_step_f64(C.byref(prm), g.nx, g.ny, it.xstart + 1, it.xstop - 1, it.ystart + 1, it.ystop - 1,
                                  *[f.device_ptr for f in cur + old + new], C.c_void_p(s.cuda_stream)))
                elif kind == "pipe": D.psy.invoke_shallow_step_dm_pipelined(prm, *cur, *old, *new, stream=s)
                else: D.psy.invoke_shallow_step_dm(prm, *cur, *old, *new, stream=s)
                old, cur, new = cur, new, old
            D.psy.halo_join(g, stream=s)
            e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) / steps
res = {}
for rep in range(2):
    for kind, skip, diag in (("plain", 0, 0), ("pipe", 0, 0), ("pipe", 1, 0), ("pipe", 3, 0), ("pipe", 3, 1), ("pipe", 3, 2), ("interior", 0, 0)):
        ms = run(kind, skip, diag); key = f"{kind}/skip{skip}/diag{diag}"; res[key] = min(ms, res.get(key, 1e9))
print(json.dumps({"tile": tile, "ms": res}, indent=1))
