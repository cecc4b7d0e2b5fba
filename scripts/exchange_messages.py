#!/usr/bin/env python3
"""What one RCCL send/recv pair costs: back-to-back halo exchanges of 1, 2, 3 fields (8 directions / 4 edges / no RCCL group)
in loop-back on one GPU -> profiles/r02_exchange_messages.txt.   python scripts/exchange_messages.py [tile]"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, dl_esm_inf_amd as D
from dm_overhead import loopback_tables
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; steps = 200
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
F = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(3)]
t = loopback_tables(D, F[0].internal); plan = C.c_void_p()
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan))); g._halo_plan = plan
s = torch.cuda.Stream()
def run(nf, skip, dirs=D._cabi.DIRS_ALL):
    L.dlesm_set_tuning(b"dm_skip_parts", skip)
    with torch.cuda.stream(s):
        for phase in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(steps if phase else 20):
                D.psy.halo_exchange_multi(F[:nf], stream=s, dirs=dirs)
            e1.record(s)
    s.synchronize()
    return round(e0.elapsed_time(e1) / steps * 1e3, 2)
res = {}
for single in (0, 1, 0, 1):
    L.dlesm_set_tuning(b"dm_aggregate_single", single)
    res[f"nf1_all8_single{single}_us_{len(res)}"] = run(1, 0)
    res[f"nf1_edges4_single{single}_us_{len(res)}"] = run(1, 0, D._cabi.DIRS_ALL | D._cabi.DIRS_NO_DIAGONALS)
L.dlesm_set_tuning(b"dm_aggregate_single", 0)
for nf in (1, 2, 3):
    res[f"nf{nf}_all8_us"] = run(nf, 0)
    res[f"nf{nf}_edges4_us"] = run(nf, 0, D._cabi.DIRS_ALL | D._cabi.DIRS_NO_DIAGONALS)
    res[f"nf{nf}_no_rccl_us"] = run(nf, 1)
print(json.dumps({"tile": tile, **res}, indent=1))
