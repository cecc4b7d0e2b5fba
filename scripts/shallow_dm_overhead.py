#!/usr/bin/env python3
"""Price the distributed shallow-water step on ONE GPU with RCCL in loop-back (rank 0 is its own eight
neighbours): plain fused step / step then grouped exchange of unew, vnew, pnew / dlesm_shallow_step_dm /
its time-loop form dlesm_shallow_step_dm_pipelined (+ one join at the end, inside the timed region).
    python scripts/shallow_dm_overhead.py [tile] [peer]      peer: the plan connected to the mailboxes"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
import dl_esm_inf_amd as D  # noqa: E402
from dm_overhead import loopback_tables  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
PEER = len(sys.argv) > 2 and sys.argv[2] == "peer"
steps = 30
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
it = F["p"].internal
t = loopback_tables(D, it)
plan = C.c_void_p()
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
g._halo_plan = plan
if PEER:
    D.psy.halo_connect_peers(g, 3)
s = torch.cuda.Stream()
prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)


def init():
    for k, n in enumerate(names):
        D.psy.hash_init(F[n], 100 + k, stream=s)
        F[n].data.add_(1.0 if n[0] == "p" else -0.5)
    D.psy.halo_exchange_multi([F["u"], F["v"], F["p"]], stream=s)


def run(kind):
    cur, old, new = [F["u"], F["v"], F["p"]], [F["uold"], F["vold"], F["pold"]], [F["unew"], F["vnew"], F["pnew"]]
    with torch.cuda.stream(s):
        init()
        for phase in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(steps if phase else 5):
                if kind == "plain":
                    D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=s)
                elif kind == "serial":
                    D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=s)
                    D.psy.halo_exchange_multi(new, stream=s)
                elif kind == "pipelined":
                    D.psy.invoke_shallow_step_dm_pipelined(prm, *cur, *old, *new, stream=s)
                else:
                    D.psy.invoke_shallow_step_dm(prm, *cur, *old, *new, stream=s)
                old, cur, new = cur, new, old
            if kind == "pipelined":
                D.psy.halo_join(g, stream=s)             # inside the timed region
            e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) / steps, cur[2].data.clone()


res = {}
for kind in ("plain", "plain", "serial", "overlapped", "pipelined", "serial", "overlapped", "pipelined", "plain"):
    ms, fin = run(kind)
    res[kind] = min(ms, res.get(kind, (1e9,))[0]), fin
w = F["p"].whole      # compare the field proper (the padding beyond `whole` accumulates the re-initialisation shifts)
cut = lambda t_: t_[w.ystart - 1:w.ystop, w.xstart - 1:w.xstop]      # noqa: E731
same = bool(torch.equal(cut(res["serial"][1]), cut(res["overlapped"][1])) and
            torch.equal(cut(res["serial"][1]), cut(res["pipelined"][1])))
out = {"tile": tile, "transport": "mailboxes" if PEER else "rccl", "ms_per_step": {k: v[0] for k, v in res.items()}, "overlapped_equals_serial_bitwise": same,
       "overlapped_over_plain": res["plain"][0] / res["overlapped"][0],
       "pipelined_over_plain": res["plain"][0] / res["pipelined"][0]}
print(json.dumps(out, indent=1))
assert same
