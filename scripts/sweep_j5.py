#!/usr/bin/env python3
"""Tuning sweep of the Jacobi-5 kernel on one GPU: every (variant, rows) pair timed in
interleaved rounds inside ONE process (guide rule 24), median and min reported.

    python scripts/sweep_j5.py [--tile 16384] [--alignment 64] [--rounds 5] [--reps 10]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=16384)
    ap.add_argument("--alignment", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--variants", type=str, default="0,1,2,3")
    ap.add_argument("--rows", type=str, default="8,16,32,64,128,256")
    ap.add_argument("--unroll", type=str, default="4")
    ap.add_argument("--out", type=str, default="gpurun_out/sweep_j5.json")
    args = ap.parse_args()
    import torch
    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    torch.cuda.set_device(0)
    os.environ["DL_ESM_ALIGNMENT"] = str(args.alignment)
    D.parallel_init(0, 1)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(args.tile, args.tile)
    D.grid_init(g, 1.0, 1.0)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    D.psy.hash_init(a, 20261004)
    D.copy_field(a, b)
    s = torch.cuda.Stream()
    combos = [(int(v), int(r), int(u)) for v in args.variants.split(",") for r in args.rows.split(",")
              for u in args.unroll.split(",")]
    times = {c: [] for c in combos}
    bytes_launch = 16.0 * args.tile * args.tile
    with torch.cuda.stream(s):
        for rnd in range(args.rounds + 1):
            for (v, r, u) in combos:
                L.dlesm_set_tuning(b"j5_variant", v)
                L.dlesm_set_tuning(b"j5_rows", r)
                L.dlesm_set_tuning(b"j5_unroll", u)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(args.reps):
                    D.psy.invoke_jacobi5(b, a, stream=s)
                    a, b = b, a
                e1.record(s)
                s.synchronize()
                if rnd > 0:                                    # round 0 is warm-up
                    times[(v, r, u)].append(e0.elapsed_time(e1) / args.reps)
    rows = []
    for (v, r, u), ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        rows.append({"variant": v, "rows": r, "unroll": u, "ms_median": med, "ms_min": mn,
                     "gbs_median": bytes_launch / med / 1e6, "gbs_best": bytes_launch / mn / 1e6})
    rows.sort(key=lambda x: x["ms_median"])
    print(f"tile {args.tile} A={args.alignment} ld={g.nx}: (variant bits: 1=nt 2=serpentine 4=VEC1)")
    for x in rows:
        print(f"  variant {x['variant']} rows {x['rows']:4d} U{x['unroll']}: median {x['ms_median']:.4f} ms "
              f"({x['gbs_median']:.0f} GB/s, {x['gbs_median'] / 80:.1f}% of 8 TB/s)  best {x['gbs_best']:.0f} GB/s")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({"tile": args.tile, "alignment": args.alignment, "ld": g.nx, "results": rows},
              open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
