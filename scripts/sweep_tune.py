#!/usr/bin/env python3
"""Generic interleaved A/B timing of Jacobi-5 tuning combinations on one GPU (one process, N
rounds, median + min; guide rule 24).

    python scripts/sweep_tune.py --grid "j5_kernel=0;j5_tile_rows=2,4,8;j5_tpb=0,4,8;j5_variant=0,1"
    python scripts/sweep_tune.py --grid "..." --grid "j5_kernel=1;j5_rows=8;j5_unroll=8;j5_variant=2"
Each --grid is a cartesian product of key=v1,v2,... terms separated by ';'.
"""
import argparse
import itertools
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEFAULTS = dict(j5xt_march=0, j5xt_march_ring=9, j5xt_march_slots=3072, j5xt_rows=0, j5xt_dpp=1, j5_autoshape=1, j5_kernel=0,j5_tile_rows=0, j5_tpb=0, j5_skew=1, j5_pad_tiles=0, j5_variant=0, j5_rows=0, j5_unroll=4)


def expand(grid):
    terms = []
    for t in grid.split(";"):
        k, vs = t.split("=")
        terms.append([(k.strip(), int(v)) for v in vs.split(",")])
    return [dict(c) for c in itertools.product(*terms)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=16384)
    ap.add_argument("--alignment", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--grid", action="append", required=True)
    ap.add_argument("--fused", type=int, default=1, help="time the fused kernel (one launch = this many steps)")
    ap.add_argument("--out", type=str, default="gpurun_out/sweep_tune.json")
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
    combos = [c for gr in args.grid for c in expand(gr)]
    times = [[] for _ in combos]
    bytes_launch = 16.0 * args.tile * args.tile
    with torch.cuda.stream(s):
        for rnd in range(args.rounds + 1):
            for idx, c in enumerate(combos):
                for k, v in {**DEFAULTS, **c}.items():
                    L.dlesm_set_tuning(k.encode(), v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(args.reps):
                    if args.fused > 1:
                        D.psy.invoke_jacobi5_multi(b, a, args.fused, stream=s)
                    else:
                        D.psy.invoke_jacobi5(b, a, stream=s)
                    a, b = b, a
                e1.record(s)
                s.synchronize()
                if rnd > 0:
                    times[idx].append(e0.elapsed_time(e1) / args.reps)
    rows = []
    for c, ts in zip(combos, times):
        med, mn = statistics.median(ts), min(ts)
        rows.append({"tuning": c, "ms_median": med, "ms_min": mn,
                     "gbs_median": bytes_launch / med / 1e6, "gbs_best": bytes_launch / mn / 1e6})
    rows.sort(key=lambda x: x["ms_median"])
    print(f"tile {args.tile} A={args.alignment} ld={g.nx}")
    for x in rows:
        tag = " ".join(f"{k[3:]}={v}" for k, v in x["tuning"].items())
        print(f"  {tag:48s} median {x['ms_median']:.4f} ms ({x['gbs_median']:.0f} GB/s, "
              f"{x['gbs_median'] / 80:.1f}%)  best {x['gbs_best']:.0f} GB/s")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({"tile": args.tile, "alignment": args.alignment, "ld": g.nx, "results": rows},
              open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
