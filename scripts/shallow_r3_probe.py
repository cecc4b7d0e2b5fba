#!/usr/bin/env python3
"""Round-3 measurements of BASELINE configs[3] (GOcean shallow water, 8192 x 8192 fp64, one MI355X), one process:

  ceilings   linear sweeps with the same stream counts as the kernels (dlesm_lab_stream_copy_f64), with / without nt
  fused      the fused NE step under the code-generation variants of sw_nt (bit 0 nt loads of the old level, bit 1 nt
             stores, bit 2 old level requested first, bit 3 straight-line form) and sw_stack (tiles stacked per group)
  periodic   the SW-offset periodic model: step + two copy launches against the one-launch form
  kernels    the seven GOcean kernels + time_smooth one by one (what a generated PSy layer launches) and in sequence

    python scripts/shallow_r3_probe.py [--tile 8192] [--steps 30] [--what ceilings,fused,periodic,kernels]
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
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--alignment", type=int, default=64, help="DL_ESM_ALIGNMENT (1 = the reference's default: odd leading dimension)")
    ap.add_argument("--what", default="ceilings,fused,periodic,kernels")
    ap.add_argument("--out", default="gpurun_out/shallow_r3_probe.json")
    args = ap.parse_args()
    what = set(args.what.split(","))
    import torch
    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    torch.cuda.set_device(0)
    os.environ["DL_ESM_ALIGNMENT"] = str(args.alignment)
    D.parallel_init(0, 1)
    s = torch.cuda.Stream()
    N = args.tile
    cells = N * N
    res = {"tile": N, "steps": args.steps}

    def tune(**kw):
        for k, v in kw.items():
            L.dlesm_set_tuning(k.encode(), v)

    def timed(fn, reps):
        """ms per call of fn(): min over passes of (reps back-to-back calls between two events on the stream)"""
        best = 1e30
        for rnd in range(args.passes + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                fn()
            e1.record(s)
            s.synchronize()
            if rnd:
                best = min(best, e0.elapsed_time(e1) / reps)
        return best

    def make_grid(sw):
        if sw:
            g = D.grid_type(D.GO_ARAKAWA_C, (0, 0, 2), D.GO_OFFSET_SW)
        else:
            g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
        g.decompose(N, N)
        D.grid_init(g, 1.0e5, 1.0e5)
        return g

    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS, "c": D.GO_U_POINTS, "z": D.GO_F_POINTS, "h": D.GO_T_POINTS}
    names9 = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]

    def make_state(g, periodic):
        F = {}
        with torch.cuda.stream(s):
            for k, name in enumerate(names9 + ["cu", "cv", "z", "h"]):
                f = D.r2d_field(g, pts[name[0]])
                if k < 9:
                    D.psy.hash_init(f, 20261004 + k, box=f.internal if periodic else None, stream=s)
                    f.data.add_(1.0 if name[0] == "p" else -0.5)
                    if periodic:
                        D.psy.apply_periodic_halos(f, stream=s)
                F[name] = f
        s.synchronize()
        return F

    g = make_grid(False)
    F = make_state(g, False)
    ld, ny = g.nx, g.ny
    nfield = (ld * ny) & ~1          # the linear sweeps take an even number of doubles
    prm = D.psy.shallow_params(g.dx, g.dy, 90.0)
    tdt = 180.0

    # ---------------------------------------------------------------- ceilings
    if "ceilings" in what:
        allf = [F[n] for n in names9]
        out = {}
        with torch.cuda.stream(s):
            for (nr, nw) in ((1, 1), (2, 1), (3, 1), (4, 1), (6, 3)):
                src = (C.c_void_p * nr)(*[f.device_ptr.value for f in allf[:nr]])
                dst = (C.c_void_p * nw)(*[f.device_ptr.value for f in allf[6:6 + nw]])
                for nt in (0, 2, 3, 1):
                    def go():
                        D._cabi.check_lab(D._cabi.lab().dlesm_lab_stream_copy_f64(nr, nw, src, dst, nfield, nt, C.c_void_p(s.cuda_stream)))
                    ms = timed(go, args.steps)
                    gbs = 8.0 * (nr + nw) * nfield / ms / 1e6
                    out[f"{nr}r+{nw}w nt={nt}"] = {"ms": ms, "gbs": gbs, "frac": gbs / 8000.0}
                    print(f"ceiling {nr}r+{nw}w nt={nt}: {ms:.4f} ms  {gbs:7.0f} GB/s  {gbs / 80:.1f} %", flush=True)
            # the initial state again (the sweeps above wrote the 'new' arrays only: nothing to restore)
        res["ceilings"] = out

    # ---------------------------------------------------------------- fused step variants
    if "fused" in what:
        out = {}
        variants = [(2, 1), (10, 1), (3, 1), (11, 1), (14, 1), (15, 1), (8, 1), (0, 1), (6, 1), (10, 2), (10, 4), (2, 2), (2, 4),
                    (11, 2), (2, 1), (10, 1)]
        with torch.cuda.stream(s):
            for plan in (False, True):
                for (nt, stack) in variants:
                    tune(sw_nt=nt, sw_stack=stack, j5_use_tuned=1 if plan else 0)
                    cur, old, new = [F[n] for n in "uvp"], [F[n + "old"] for n in "uvp"], [F[n + "new"] for n in "uvp"]
                    if plan:
                        if stack != 1 or nt not in (2, 10, 11):
                            continue
                        D.psy.autotune_shallow(prm, *cur, *old, *new, stream=s)

                    def go():
                        D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=s)
                    ms = timed(go, args.steps)
                    gbs = 72.0 * cells / ms / 1e6
                    key = f"sw_nt={nt} stack={stack}" + (" planned" if plan else "")
                    k2, n = key, 1
                    while k2 in out:
                        n += 1
                        k2 = f"{key} #{n}"
                    out[k2] = {"ms": ms, "gbs": gbs, "frac": gbs / 8000.0}
                    print(f"fused {k2:34s}: {ms:.4f} ms  {gbs:7.0f} GB/s  {gbs / 80:.2f} %", flush=True)
            tune(sw_nt=2, sw_stack=1, j5_use_tuned=1)
        res["fused"] = out

    # ---------------------------------------------------------------- the seven kernels one by one
    if "kernels" in what:
        out = {}
        xs, xe, ys, ye = F["p"].internal.box()
        off = g.offset
        sp = C.c_void_p(s.cuda_stream)
        P = {n: F[n].device_ptr for n in F}
        calls = {
            "cu": (24, lambda: L.dlesm_compute_cu_f64(off, ld, ny, xs - 1, xe, ys, ye + 1, P["cu"], P["p"], P["u"], sp)),
            "cv": (24, lambda: L.dlesm_compute_cv_f64(off, ld, ny, xs, xe + 1, ys - 1, ye, P["cv"], P["p"], P["v"], sp)),
            "z": (32, lambda: L.dlesm_compute_z_f64(off, ld, ny, xs - 1, xe, ys - 1, ye, prm.fsdx, prm.fsdy, P["z"], P["p"], P["u"], P["v"], sp)),
            "h": (32, lambda: L.dlesm_compute_h_f64(off, ld, ny, xs, xe + 1, ys, ye + 1, P["h"], P["p"], P["u"], P["v"], sp)),
            "unew": (40, lambda: L.dlesm_compute_unew_f64(off, ld, ny, xs, xe, ys, ye, prm.tdts8, prm.tdtsdx, P["unew"], P["uold"], P["z"], P["cv"], P["h"], sp)),
            "vnew": (40, lambda: L.dlesm_compute_vnew_f64(off, ld, ny, xs, xe, ys, ye, prm.tdts8, prm.tdtsdy, P["vnew"], P["vold"], P["z"], P["cu"], P["h"], sp)),
            "pnew": (32, lambda: L.dlesm_compute_pnew_f64(off, ld, ny, xs, xe, ys, ye, prm.tdtsdx, prm.tdtsdy, P["pnew"], P["pold"], P["cu"], P["cv"], sp)),
            "time_smooth": (32, lambda: L.dlesm_time_smooth_f64(ld, ny, xs, xe, ys, ye, 0.001, P["u"], P["unew"], P["uold"], sp)),
        }
        with torch.cuda.stream(s):
          for swk_nt in (0, 1, -1, 10, 11, 9):      # >= 9: + non-temporal loads of the once-read arrays (swk_ntl), store policy swk_nt - 10
            tune(swk_nt=swk_nt if swk_nt < 9 else swk_nt - 10, swk_ntl=1 if swk_nt >= 9 else (0 if swk_nt >= 0 else -1))
            total = 0.0
            for name, (bytes_per_cell, fn) in calls.items():
                def go():
                    D._cabi.check(fn())
                ms = timed(go, args.steps)
                gbs = bytes_per_cell * cells / ms / 1e6
                out[name + ("" if swk_nt < 0 else f" swk_nt={swk_nt}")] = {"ms": ms, "bytes_per_cell": bytes_per_cell, "gbs": gbs, "frac": gbs / 8000.0}
                if name != "time_smooth":
                    total += ms
                print(f"kernel {name:12s} swk_nt={swk_nt:2d}: {ms:.4f} ms  {bytes_per_cell} B/cell  {gbs:7.0f} GB/s  {gbs / 80:.2f} %", flush=True)

            if "shapes" in what:      # launch-shape landscape of each kernel: waves per group x padding tiles per row
                tune(j5_autoshape=0)
                for name, (bytes_per_cell, fn) in calls.items():
                    line = []
                    for tpb in (4, 8):
                        for pad in range(0, 6):
                            tune(j5_tpb=tpb, j5_pad_tiles=pad)
                            ms = timed(lambda: D._cabi.check(fn()), 10)
                            line.append(f"{tpb}/{65 + pad}:{bytes_per_cell * cells / ms / 1e6 / 80:.1f}")
                            out[f"shape {name} tpb={tpb} nxw={65 + pad}"] = ms
                    print(f"shapes {name:12s} " + " ".join(line), flush=True)
                tune(j5_autoshape=1, j5_tpb=0, j5_pad_tiles=0)
            tune(swk_nt=-1, swk_ntl=-1)

            def seq():   # (default store policy: the last pass above)
                D.psy.invoke_shallow_kernel_sequence(tdt, *[F[n] for n in names9[:6]], F["cu"], F["cv"], F["z"], F["h"],
                                                     F["unew"], F["vnew"], F["pnew"], stream=s)
            ms = timed(seq, max(4, args.steps // 4))
            gbs = 224.0 * cells / ms / 1e6
            out["sequence of 7"] = {"ms": ms, "sum_of_kernels_ms": total, "bytes_per_cell": 224, "gbs": gbs, "frac": gbs / 8000.0,
                                    "mcells_per_s": cells / ms / 1e3}
            print(f"sequence of 7 launches: {ms:.4f} ms (sum of the kernels {total:.4f})  {gbs:7.0f} GB/s  {gbs / 80:.2f} %  "
                  f"{cells / ms / 1e3:.0f} Mcells/s", flush=True)
        res["kernels"] = out

    # ---------------------------------------------------------------- SW-offset periodic model
    if "periodic" in what:
        del F
        torch.cuda.empty_cache()
        gs = make_grid(True)
        G = make_state(gs, True)
        out = {}
        with torch.cuda.stream(s):
            for nt in (2, 10, 11):
                tune(sw_nt=nt)
                cur, old, new = [G[n] for n in "uvp"], [G[n + "old"] for n in "uvp"], [G[n + "new"] for n in "uvp"]

                def three():
                    D.psy.invoke_shallow_step_sw(prm, *cur, *old, *new, stream=s)
                    D.psy.apply_periodic_halos_multi(new, stream=s)

                def one():
                    D.psy.invoke_shallow_step_sw_periodic(prm, *cur, *old, *new, stream=s)

                def plain():
                    D.psy.invoke_shallow_step_sw(prm, *cur, *old, *new, stream=s)
                for label, fn in (("step only", plain), ("step + 2 copy launches", three), ("one launch", one), ("step only #2", plain),
                                  ("one launch #2", one)):
                    ms = timed(fn, args.steps)
                    gbs = 72.0 * cells / ms / 1e6
                    out[f"sw_nt={nt} {label}"] = {"ms": ms, "gbs": gbs, "frac": gbs / 8000.0}
                    print(f"periodic sw_nt={nt} {label:24s}: {ms:.4f} ms  {gbs:7.0f} GB/s  {gbs / 80:.2f} %", flush=True)
            tune(sw_nt=2)
            # the seven kernels on this (SW-offset, periodic) grid: with the periodic copies of the intermediates, and bare
            cur, old, new = [G[n] for n in "uvp"], [G[n + "old"] for n in "uvp"], [G[n + "new"] for n in "uvp"]

            def seq_p():
                D.psy.invoke_shallow_kernel_sequence(tdt, *cur, *old, G["cu"], G["cv"], G["z"], G["h"], *new, stream=s)

            def seq_p_halos():
                seq_p()
                D.psy.apply_periodic_halos_multi(new, stream=s)

            def copies4():
                D.psy.apply_periodic_halos_multi([G["cu"], G["cv"], G["z"], G["h"]], stream=s)
            for label, fn in (("seven kernels + copies of cu, cv, z, h", seq_p), ("... + copies of the new level", seq_p_halos),
                              ("the periodic copies of four fields alone (2 launches)", copies4)):
                ms = timed(fn, max(4, args.steps // 3))
                out[label] = {"ms": ms}
                print(f"periodic {label:56s}: {ms:.4f} ms  {cells / ms / 1e3:9.0f} Mcells/s", flush=True)
        res["periodic"] = out

    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
