#!/usr/bin/env python3
"""Generate tests/golden/jacobi5_64.json from the CPU ORACLE (oracle/dlesm_oracle.c).

The reference contains no stencil loop, so these vectors pin the HIP
kernels to the oracle's arithmetic, not to the reference: "parity unpinned" in the sense of
DESIGN.md section 3.  Checksums are long-double SUM(ABS()) of the internal region.

    python tests/golden/make_stencil_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SEED = 20261004


def main():
    n = 64
    d, subs = O.decompose(n, n, 1)
    nx, ny = O.grid_extents(subs[0].glob.nx, subs[0].glob.ny)          # 67 x 67, no alignment
    a = O.hash_field(SEED, ny, nx, 0, 0, 1, n + 2, 1, n + 2)           # whole region incl. ring; local cell 1 = global cell 0
    b = a.copy()
    cs = {}
    for step in range(1, 11):
        O.jacobi5(a, b, nx, 2, n + 1, 2, n + 1)
        a, b = b, a
        if step in (1, 5, 10):
            cs[str(step)] = O.lib().orc_checksum(a, nx, 2, n + 1, 2, n + 1)
    samples = [[j, i, float(a[j - 1, i - 1])] for (j, i) in
               [(2, 2), (2, 65), (65, 2), (65, 65), (33, 17), (10, 60), (1, 1), (66, 66)]]
    out = {"_provenance": "oracle/dlesm_oracle.c orc_jacobi5, 64x64 interior, hash init seed "
                          f"{SEED} on the whole region, 10 ping-pong steps; NOT a reference output",
           "n": n, "nx": nx, "ny": ny, "seed": SEED, "checksums": cs, "samples": samples}
    with open(os.path.join(HERE, "jacobi5_64.json"), "w") as f:
        json.dump(out, f)
        f.write("\n")
    print(out)


if __name__ == "__main__":
    main()
