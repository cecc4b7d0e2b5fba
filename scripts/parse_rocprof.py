#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into the small files committed under profiles/.

    python scripts/parse_rocprof.py stats  <dir> <out.md>     # --kernel-trace --stats run
    python scripts/parse_rocprof.py pmc    <fetch_dir> <write_dir> <key> <out.json> [kernel-substring]

pmc mode applies the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md section HBM:
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B read requests as 64 B for wide
coalesced streams, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import statistics
import sys


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits


def stats(d, out):
    rows = []
    for f in find(d, "kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    trace = []
    for f in find(d, "kernel_trace.csv"):
        trace += list(csv.DictReader(open(f)))
    lines = ["# rocprofv3 --kernel-trace --stats summary", "", f"source dir: `{d}`", ""]
    if rows:
        keys = [k for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")
                if k in rows[0]]
        lines += ["| " + " | ".join(keys) + " |", "|" + "---|" * len(keys)]
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        for r in rows[:15]:
            lines.append("| " + " | ".join(str(r[k])[:90] for k in keys) + " |")
    if trace:
        by = {}
        for r in trace:
            name = r.get("Kernel_Name", "?")
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            # one row per (kernel, launch geometry): a bench run launches the same kernel on several tile sizes
            key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
            by.setdefault(key, []).append((dur, r))
        lines += ["", "## per-kernel durations from the dispatch trace (ns), one row per launch geometry", "",
                  "| kernel | launches | mean | median | min | max | VGPR | LDS | grid (threads) | workgroup |", "|---|---|---|---|---|---|---|---|---|---|"]
        for (name, gx, wx), v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:24]:
            ds = [x[0] for x in v]
            r = v[0][1]
            lines.append(f"| {name[:80]} | {len(ds)} | {statistics.mean(ds):.0f} | {statistics.median(ds):.0f} | "
                         f"{min(ds):.0f} | {max(ds):.0f} | {r.get('VGPR_Count', '')} | {r.get('LDS_Block_Size', '')} | "
                         f"{gx} | {wx} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def counter_rows(d, counter, kernel_sub):
    vals = []
    for f in find(d, "counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and kernel_sub in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    return vals


def pmc(fetch_dir, write_dir, key, out, kernel_sub="jacobi5_", last="0"):
    """last = N: only the last N launches of the kernel (the warm-up + timed region of bench.py, after
    the planning call has tried its launch shapes)"""
    fetch = counter_rows(fetch_dir, "FETCH_SIZE", kernel_sub)
    write = counter_rows(write_dir, "WRITE_SIZE", kernel_sub)
    if not fetch or not write:
        sys.exit(f"no counter rows for {kernel_sub}: fetch {len(fetch)} write {len(write)}")
    if int(last) > 0:
        fetch, write = fetch[-int(last):], write[-int(last):]
    f_kib, w_kib = statistics.median(fetch), statistics.median(write)
    rec = {
        "kernel": kernel_sub, "launches_seen": [len(fetch), len(write)],
        "FETCH_SIZE_KiB_median": f_kib, "WRITE_SIZE_KiB_median": w_kib,
        "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950: 128-B requests tallied at 64 B); "
                      "write bytes = WRITE_SIZE x 1024",
        "hbm_read_bytes_per_launch": 2.0 * f_kib * 1024.0,
        "hbm_write_bytes_per_launch": w_kib * 1024.0,
        "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
    }
    allrec = json.load(open(out)) if os.path.exists(out) else {}
    allrec[key] = rec
    json.dump(allrec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:])
