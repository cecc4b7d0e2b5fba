#!/usr/bin/env python3
"""Price the distributed Jacobi step on ONE GPU: RCCL send/recv in loop-back (rank 0 is its own
west/east/south/north/corner neighbour, i.e. a periodic wrap), so that the whole machinery of
dlesm_jacobi5_step_dm -- frame kernel, side-stream pack + grouped ncclSend/ncclRecv + unpack,
interior kernel, event join -- runs exactly as it does on 8 GPUs, minus the xGMI hop.
Reports ms/step of (a) the plain single-tile step, (b) exchange-then-step without overlap,
(c) the overlapped distributed step, and checks (b) == (c) bit for bit.

    python scripts/dm_overhead.py [--tile 8192] [--steps 50]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def loopback_tables(D, it, d=1):
    """tables of a depth-d exchange in which rank 0 is its own eight neighbours"""
    t = D._cabi.CommTables()
    xl, xh, yl, yh = it.xstart, it.xstop - d + 1, it.ystart, it.ystop - d + 1   # low / high strips
    msgs = [  # dir, isrc, jsrc, ides, jdes, nx, ny   (a periodic wrap onto oneself)
        (2, xh, yl, it.xstart - d, yl, d, it.ny),
        (1, xl, yl, it.xstop + 1, yl, d, it.ny),
        (4, xl, yh, xl, it.ystart - d, it.nx, d),
        (3, xl, yl, xl, it.ystop + 1, it.nx, d),
        (6, xh, yh, it.xstart - d, it.ystart - d, d, d),
        (5, xl, yl, it.xstop + 1, it.ystop + 1, d, d),
        (7, xl, yh, it.xstop + 1, it.ystart - d, d, d),
        (8, xh, yl, it.xstart - d, it.ystop + 1, d, d),
    ]
    t.nsend = t.nrecv = len(msgs)
    for k, (d_, isrc, jsrc, ides, jdes, nx, ny) in enumerate(msgs):
        t.dirsend[k] = t.dirrecv[k] = d_
        t.destination[k] = t.source[k] = 0
        t.isrcsend[k], t.jsrcsend[k], t.idessend[k], t.jdessend[k] = isrc, jsrc, ides, jdes
        t.nxsend[k], t.nysend[k] = nx, ny
        t.isrcrecv[k], t.jsrcrecv[k], t.idesrecv[k], t.jdesrecv[k] = isrc, jsrc, ides, jdes
        t.nxrecv[k], t.nyrecv[k] = nx, ny
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--out", default="gpurun_out/dm_overhead.json")
    ap.add_argument("--fused", type=int, default=1, help="T > 1: the fused T-step forms, depth-T halos")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT", help="dlesm_set_tuning before init")
    ap.add_argument("--peer", action="store_true", help="connect the plan's mailboxes: the distributed steps use the peer "
                    "transport (frame workgroups store into the neighbour's mailbox; no RCCL kernel)")
    args = ap.parse_args()
    import torch
    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    for kv in args.tune:
        k, v = kv.split("=")
        L.dlesm_set_tuning(k.encode(), int(v))
    torch.cuda.set_device(0)
    os.environ["DL_ESM_ALIGNMENT"] = "64"
    D.parallel_init(0, 1, use_rccl=True)
    T = args.fused
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(args.tile, args.tile, halo_width=T)
    D.grid_init(g, 1.0, 1.0)
    F = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(2)]
    it = F[0].internal
    tables = loopback_tables(D, it, T)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(tables), g.nx, g.ny, C.byref(plan)))
    if args.peer:
        D._cabi.check(L.dlesm_halo_plan_peer_connect_rccl(plan, 1))
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    box = it.box()
    ebox = (box[0] - 1, box[1] + 1, box[2] - 1, box[3] + 1)    # last stage box of a tile with 8 neighbours

    def init(a, b):
        D.psy.hash_init(a, 20261004, stream=s)
        D._cabi.check(L.dlesm_halo_exchange_f64(plan, a.device_ptr, D._cabi.DIRS_ALL, sp))
        D.copy_field(a, b, stream=s)

    def plain(a, b):
        if T > 1:       # the whole tile in one fused launch, no exchange
            D._cabi.check(L.dlesm_stencil5_multi_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, T, *box, *ebox,
                                                     1, 1, 1, 1, sp))
        else:
            D._cabi.check(L.dlesm_stencil5_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, *box, sp))

    # the single-step distributed form exchanges the four edges only (a 5-point stencil reads no corner)
    EXMASK = D._cabi.DIRS_ALL if T > 1 else D._cabi.DIRS_EDGES_ONLY

    def serial(a, b):   # stencil, then the exchange of the result on the same stream: no overlap
        plain(a, b)
        D._cabi.check(L.dlesm_halo_exchange_f64(plan, b.device_ptr, EXMASK, sp))

    def overlapped(a, b):
        if T > 1:
            D._cabi.check(L.dlesm_jacobi5_multi_step_dm(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, T, *box, sp))
        else:
            D._cabi.check(L.dlesm_jacobi5_step_dm(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *box, sp))

    def pipelined(a, b):    # the time-loop form: no join on the caller's stream between steps
        D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *box, sp))

    res = {}
    finals = {}
    with torch.cuda.stream(s):
        # the first mode of a process runs on cold clocks (2-3 % slow): measure `plain` twice, keep the second;
        # then serial / overlapped twice interleaved, keep the faster of each
        modes = [("cold", plain), ("plain", plain), ("serial", serial), ("overlapped", overlapped),
                 ("serial2", serial), ("overlapped2", overlapped), ("plain2", plain)]
        if T == 1:
            modes += [("pipelined", pipelined), ("pipelined2", pipelined), ("plain3", plain)]
        for name, fn in modes:
            a, b = F[0], F[1]                         # the same two buffers for every mode
            init(a, b)
            for _ in range(5):
                fn(a, b)
                a, b = b, a
            init(a, b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(args.steps):
                fn(a, b)
                a, b = b, a
            if name.startswith("pipelined"):
                D._cabi.check(L.dlesm_halo_plan_join(plan, sp))     # the one join of the loop, inside the timed region
            e1.record(s)
            s.synchronize()
            res[name] = e0.elapsed_time(e1) / args.steps
            if name in ("serial", "overlapped", "pipelined"):
                finals[name] = a.data.clone()
    for k in ("plain", "serial", "overlapped") + (("pipelined",) if T == 1 else ()):
        res[k] = min(res[k], res.pop(k + "2"))
    if T == 1:
        res["plain"] = min(res["plain"], res.pop("plain3"))
    res.pop("cold")
    same = bool(torch.equal(finals["serial"], finals["overlapped"]))
    if T == 1:
        same = same and bool(torch.equal(finals["serial"], finals["pipelined"]))
    cells = args.tile * args.tile
    out = {"tile": args.tile, "tuning": args.tune, "launches": args.steps, "time_steps_per_launch": T, "ms_per_launch": res,
           "mcells_per_s": {k: cells * T / v / 1e3 for k, v in res.items()},
           "overlapped_equals_serial_bitwise": same,
           "overlap_efficiency_vs_plain": res["plain"] / res["overlapped"]}
    if T == 1:
        out["pipelined_efficiency_vs_plain"] = res["plain"] / res["pipelined"]
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    assert same, "overlapped distributed step differs from stencil+exchange"


if __name__ == "__main__":
    main()
