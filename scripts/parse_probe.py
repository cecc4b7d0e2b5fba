#!/usr/bin/env python3
"""tabulate a pmc_probe.py run: per dispatch (in order) kernel, counter, value in bytes"""
import csv, glob, os, sys
rows = []
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((os.path.basename(d.rstrip("/")), int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r["Counter_Name"],
                         float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
for r in rows:
    print(f"{r[0]:12s} #{r[1]:3d} {r[2]:60s} {r[3]:22s} {r[4]:16.1f}  {r[5]/1e3:9.1f} us")
