#!/usr/bin/env python3
"""r2d_field%halo_exchange, back to back, loop-back on one GPU: the RCCL group form against the mailbox form
(dlesm_halo_plan_peer_connect; two small launches, no RCCL), 1-3 fields, eight directions / four edges, us per exchange
-> profiles/r03_exchange_peer.txt.     python scripts/exchange_peer_bench.py [tile]"""
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
steps = 200
L = D._cabi.lib()
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(tile, tile)
D.grid_init(g, 1.0, 1.0)
F = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(3)]
t = loopback_tables(D, F[0].internal)
plan = C.c_void_p()
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
g._halo_plan = plan
D.psy.halo_connect_peers(g, 3)
s = torch.cuda.Stream()
for k, f in enumerate(F):
    D.psy.hash_init(f, 7 + k, stream=s)


def run(nf, peer, dirs):
    L.dlesm_set_tuning(b"dm_peer_exchange", peer)
    best = 1e9
    with torch.cuda.stream(s):
        for phase in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(steps if phase else 20):
                D.psy.halo_exchange_multi(F[:nf], stream=s, dirs=dirs)
            e1.record(s)
            s.synchronize()
            if phase:
                best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    return round(best, 2)


res = {"tile": tile}
for nf in (1, 2, 3):
    for name, dirs in (("all8", D._cabi.DIRS_ALL), ("edges4", D._cabi.DIRS_ALL | D._cabi.DIRS_NO_DIAGONALS)):
        r, p = run(nf, 0, dirs), run(nf, 1, dirs)
        res[f"nf{nf}_{name}"] = {"rccl_us": r, "mailboxes_us": p, "ratio": round(r / p, 2)}
        print(f"{tile}^2  {nf} field(s), {name:6s}: RCCL group {r:7.2f} us   mailboxes {p:6.2f} us   x{r / p:.1f}", flush=True)
A = [f.data.clone() for f in F]
L.dlesm_set_tuning(b"dm_peer_exchange", 0)
D.psy.halo_exchange_multi(F, stream=s)
s.synchronize()
same = all(bool(torch.equal(a, f.data)) for a, f in zip(A, F))
print("halos identical through both transports:", same)
res["identical"] = same
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/exchange_peer_{tile}.json", "w"), indent=1)
assert same
