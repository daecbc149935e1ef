#!/usr/bin/env python3
"""print start-to-start gaps and durations (us) of the last N dispatches of a kernel from a rocprofv3 kernel-trace CSV"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = rows[-n:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"dur {(e - s) / 1e3:7.2f} us   start-to-start {((s - prev) / 1e3) if prev else float('nan'):8.2f} us   idle before {((s - pe) / 1e3) if prev else float('nan'):8.2f} us")
    prev, pe = s, e
