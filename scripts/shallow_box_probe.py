#!/usr/bin/env python3
"""The shallow-water sweep over boxes shifted by 0-2 columns / rows against the whole box (tile anchoring).
   python scripts/shallow_box_probe.py [tile]"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, dl_esm_inf_amd as D
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; steps = 40
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=False)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
s = torch.cuda.Stream(); prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
it = F["p"].internal
def run(dx0, dx1, dy0, dy1):
    cur, old, new = [F[n] for n in names[:3]], [F[n] for n in names[3:6]], [F[n] for n in names[6:]]
    with torch.cuda.stream(s):
        for n in names:
            D.set_field(F[n], 1.0 if n[0] == "p" else 0.0, stream=s)
        for phase in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(steps if phase else 8):
                D._cabi.check(L.dlesm_shallow_step_f64(C.byref(prm), g.nx, g.ny, it.xstart + dx0, it.xstop - dx1, it.ystart + dy0, it.ystop - dy1,
                              *[f.device_ptr for f in cur + old + new], C.c_void_p(s.cuda_stream)))
                old, cur, new = cur, new, old
            e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) / steps
res = {}
cases = [(0,0,0,0), (1,1,1,1), (1,0,0,0), (0,1,0,0), (0,0,1,0), (0,0,0,1), (1,1,0,0), (0,0,1,1), (2,2,2,2), (2,0,0,0)]
for rep in range(3):
    for c in cases:
        ms = run(*c); res[str(c)] = min(ms, res.get(str(c), 1e9))
print(json.dumps({"tile": tile, "ld": g.nx, "ms": {k: round(v, 4) for k, v in res.items()}}, indent=1))
