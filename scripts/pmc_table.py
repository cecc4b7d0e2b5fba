#!/usr/bin/env python3
"""Table of per-kernel medians from a set of single-counter rocprofv3 --pmc runs.

    python scripts/pmc_table.py <dir-with-one-subdir-per-run> <kernel substring>[@<grid size>] [...]
Every sub-directory holds the output of one `rocprofv3 --pmc <COUNTER> -d <subdir>` run.  `name@grid` keeps only
the launches of that grid size (threads), i.e. one launch geometry of a kernel that a run launches at several.
"""
import csv
import glob
import os
import statistics
import sys


def main():
    root, subs = sys.argv[1], sys.argv[2:]
    table = {}
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            for sub in subs:
                name, _, grid = sub.partition("@")
                if name in r.get("Kernel_Name", "") and (not grid or r.get("Grid_Size") == grid):
                    table.setdefault((r["Counter_Name"], sub), []).append(
                        (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                         r["VGPR_Count"], r["Workgroup_Size"], r["Grid_Size"]))
    counters = sorted({k[0] for k in table})
    print(f"{'counter':34s}" + "".join(f"{s:>24s}" for s in subs))
    for c in counters:
        row = f"{c:34s}"
        for s in subs:
            v = table.get((c, s))
            row += f"{statistics.median(x[0] for x in v):24.4g}" if v else f"{'-':>24s}"
        print(row)
    for s in subs:
        anyv = next((v for (c, ss), v in table.items() if ss == s), None)
        if anyv:
            print(f"{s}: launches {len(anyv)}, median ns under the profiler {statistics.median(x[1] for x in anyv):.0f}, "
                  f"vgprs {anyv[0][2]}, workgroup {anyv[0][3]}, grid {anyv[0][4]}")


if __name__ == "__main__":
    main()
