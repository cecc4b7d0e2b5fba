#!/usr/bin/env python3
"""BASELINE config 4: GOcean shallow-water u/v/h update, 8192x8192 fp64, one MI355X.
Times the fused step (72 B/cell algorithmic) for the register-tiled kernel (R = 1, 2) and the
direct-load kernel, leapfrog buffer rotation between steps.

    python scripts/shallow_bench.py [--tile 8192] [--steps 40]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--out", default="gpurun_out/shallow_bench.json")
    ap.add_argument("--no-cpu", action="store_true", help="accepted and ignored (the CPU leg lives in bench.py)")
    ap.add_argument("--only-default", action="store_true", help="time the default kernel only (profiling runs)")
    args = ap.parse_args()
    import torch
    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    torch.cuda.set_device(0)
    os.environ["DL_ESM_ALIGNMENT"] = "64"
    D.parallel_init(0, 1)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(args.tile, args.tile)
    D.grid_init(g, 1.0, 1.0)
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    s = torch.cuda.Stream()
    F = {}
    with torch.cuda.stream(s):
        for k, name in enumerate(["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]):
            f = D.r2d_field(g, pts[name[0]])
            D.psy.hash_init(f, 20261004 + k, stream=s)
            if name[0] == "p":
                f.data.add_(1.0)
            else:
                f.data.sub_(0.5)
            F[name] = f
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
    cells = args.tile * args.tile
    res = {}
    with torch.cuda.stream(s):
        for label, kern, rows, nt in (("tile R=2", 0, 2, 0), ("tile R=2 nt old loads", 0, 2, 1), ("tile R=2 nt stores", 0, 2, 2),
                                      ("tile R=2 nt both", 0, 2, 3), ("tile R=3 nt both", 0, 3, 3), ("tile R=1", 0, 1, 0),
                                      ("tile R=3", 0, 3, 0), ("direct", 1, 2, 0), ("tile R=2 again", 0, 2, 0),
                                      ("tile R=2 nt both again", 0, 2, 3), ("tile R=2 default + planning call", 0, 2, -1),
                                      ("tile R=2 default, rule", 0, 2, -2)):
            if args.only_default and label != "tile R=2 default, rule":
                continue
            L.dlesm_set_tuning(b"sw_kernel", kern)
            L.dlesm_set_tuning(b"sw_tile_rows", rows)
            L.dlesm_set_tuning(b"sw_nt", nt if nt >= 0 else 2)
            L.dlesm_set_tuning(b"j5_use_tuned", 0 if nt == -2 else 1)
            if nt == -1:
                D.psy.autotune_shallow(prm, F["u"], F["v"], F["p"], F["uold"], F["vold"], F["pold"], F["unew"], F["vnew"],
                                       F["pnew"], stream=s)
            cur = [F["u"], F["v"], F["p"]]
            old = [F["uold"], F["vold"], F["pold"]]
            new = [F["unew"], F["vnew"], F["pnew"]]
            ts = []
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(args.steps):
                    D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=s)
                    old, cur, new = cur, new, old          # leapfrog rotation
                e1.record(s)
                s.synchronize()
                if rnd:
                    ts.append(e0.elapsed_time(e1) / args.steps)
            ms = min(ts)
            res[label] = {"ms_per_step": ms, "mcells_per_s": cells / ms / 1e3, "gbs_72B": 72.0 * cells / ms / 1e6,
                          "frac_of_8TBs": 72.0 * cells / ms / 1e6 / 8000.0}
            print(f"{label:16s} {ms:.4f} ms/step  {cells / ms / 1e3:9.0f} Mcells/s  {72.0 * cells / ms / 1e6:6.0f} GB/s "
                  f"({72.0 * cells / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
    # the SW-offset, doubly periodic form (the GOcean `shallow` benchmark's configuration): step + the
    # periodic copies of the three new fields (two launches) + rotation
    if not args.only_default:
        L.dlesm_set_tuning(b"sw_tile_rows", 2)
        L.dlesm_set_tuning(b"sw_nt", 2)
        gs = D.grid_type(D.GO_ARAKAWA_C, (0, 0, 2), D.GO_OFFSET_SW)
        gs.decompose(args.tile, args.tile)
        D.grid_init(gs, 1.0, 1.0)
        with torch.cuda.stream(s):
            G = {}
            for k, name in enumerate(["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]):
                f = D.r2d_field(gs, pts[name[0]])
                D.psy.hash_init(f, 20261004 + k, box=f.internal, stream=s)
                f.data.add_(1.0 if name[0] == "p" else -0.5)
                D.psy.apply_periodic_halos(f, stream=s)
                G[name] = f
            for label, kern in (("SW periodic, tile kernel", 0), ("SW periodic, direct kernel", 1)):
                L.dlesm_set_tuning(b"sw_kernel", kern)
                cur, old, new = [G["u"], G["v"], G["p"]], [G["uold"], G["vold"], G["pold"]], [G["unew"], G["vnew"], G["pnew"]]
                ts = []
                for rnd in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(args.steps):
                        D.psy.invoke_shallow_step_sw(prm, *cur, *old, *new, stream=s)
                        D.psy.apply_periodic_halos_multi(new, stream=s)
                        old, cur, new = cur, new, old
                    e1.record(s)
                    s.synchronize()
                    if rnd:
                        ts.append(e0.elapsed_time(e1) / args.steps)
                ms = min(ts)
                res[label] = {"ms_per_step": ms, "mcells_per_s": cells / ms / 1e3, "gbs_72B": 72.0 * cells / ms / 1e6,
                              "frac_of_8TBs": 72.0 * cells / ms / 1e6 / 8000.0}
                print(f"{label:32s} {ms:.4f} ms/step  {cells / ms / 1e3:9.0f} Mcells/s  {72.0 * cells / ms / 1e6:6.0f} GB/s "
                      f"({72.0 * cells / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
            L.dlesm_set_tuning(b"sw_kernel", 0)
    out = {"tile": args.tile, "ld": g.nx, "steps": args.steps, "algorithmic_bytes_per_cell": 72, "gpu": res}
    # (the CPU baseline of this configuration is bench.py's `shallow_water.cpu_baseline` leg: only tests/, smoke() and that leg
    #  may use the oracle)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
