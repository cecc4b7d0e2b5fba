#!/usr/bin/env python3
"""Generate tests/golden/sw_numpy_64x48.json (and the SW-offset periodic twin) from the INDEPENDENT
numpy evaluation of the shallow-water update (tests/sw_numpy.py) -- not from the oracle, not from the
HIP kernels.  The reference has no stencil loop: these vectors pin oracle and
kernels to DESIGN.md section 6 as evaluated by a second, separately written code path.

    python tests/golden/make_sw_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402   (only its numpy hash twin and the extents helper are used)
import sw_numpy as N  # noqa: E402

SEED = 20261004
POINTS = [(2, 2), (65, 2), (2, 49), (65, 49), (33, 25), (10, 40), (64, 3), (3, 48),
          (17, 17), (40, 9), (50, 30), (21, 44)]          # (i, j), 1-based


def main():
    nx_i, ny_i = 64, 48
    d, subs = O.decompose(nx_i, ny_i, 1)
    ld, ny_arr = O.grid_extents(subs[0].glob.nx, subs[0].glob.ny)       # 67 x 51, DL_ESM_ALIGNMENT unset
    box = (2, nx_i + 1, 2, ny_i + 1)
    whole = (1, nx_i + 2, 1, ny_i + 2)
    prm = N.Params(1.0e5, 1.0e5, 90.0)
    cur, old, new = N.initial_state(O.hash_field, SEED, ny_arr, ld, whole)
    rec = {}

    def step(c, o, n):
        N.sw_step_numpy(prm, box, *c, *o, *n)

    def on_step(k, c):
        if k in (1, 5, 10):
            rec[str(k)] = {
                name: {"abs_sum": N.abs_sum(f, box).hex(), "sha256": N.digest(f, box),
                       "samples": [[i, j, float(f[j - 1, i - 1]).hex()] for (i, j) in POINTS]}
                for name, f in zip("uvp", c)}

    N.leapfrog(step, 10, cur, old, new, on_step)
    out = {"_provenance": "tests/sw_numpy.py sw_step_numpy (whole-array numpy evaluation of DESIGN.md section 6, "
                          f"NE offset), {nx_i}x{ny_i} interior, ld {ld}, hash init seeds {SEED}+0/1/2 on the whole "
                          "region (u,v - 0.5; p + 1.0 on the complete array), old/new levels start as copies, leapfrog "
                          "by rotation, dx=dy=1e5, dt=90; floats are C99 hex; NOT a reference output",
           "nx": nx_i, "ny": ny_i, "ld": ld, "ny_arr": ny_arr, "seed": SEED, "dx": 1.0e5, "dy": 1.0e5, "dt": 90.0,
           "steps": rec}
    with open(os.path.join(HERE, "sw_numpy_64x48.json"), "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print("wrote sw_numpy_64x48.json:", {k: v["p"]["abs_sum"] for k, v in rec.items()})


if __name__ == "__main__":
    main()
