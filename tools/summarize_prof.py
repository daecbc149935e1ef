#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/...) into the small summaries committed under profiles/.
usage: summarize_prof.py <kernel_stats.csv> [<counter_collection.csv> ...] > profiles/<name>.txt"""
import collections
import csv
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:70]


def main():
    stats = sys.argv[1]
    print(f"# rocprofv3 --kernel-trace --stats  ({stats})")
    print(f"{'kernel':72s} {'calls':>6s} {'avg_ns':>10s} {'min_ns':>8s} {'max_ns':>8s} {'pct':>6s}")
    for r in csv.DictReader(open(stats)):
        print(f"{short(r['Name']):72s} {r['Calls']:>6s} {float(r['AverageNs']):10.1f} {r['MinNs']:>8s} {r['MaxNs']:>8s} {r['Percentage']:>6s}")
    for f in sys.argv[2:]:
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            vals[(r["Counter_Name"], short(r["Kernel_Name"]), r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        print(f"\n# rocprofv3 --pmc  ({f})   FETCH_SIZE / WRITE_SIZE are in KiB per dispatch")
        print(f"{'counter':12s} {'kernel':60s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'wg':>5s} {'grid':>7s} {'n':>4s} {'mean':>10s} {'min':>10s} {'max':>10s}")
        for (c, k, vg, sg, lds, wg, grid), v in vals.items():
            print(f"{c:12s} {k[:60]:60s} {vg:>5s} {sg:>5s} {lds:>7s} {wg:>5s} {grid:>7s} {len(v):4d} {sum(v)/len(v):10.3f} {min(v):10.3f} {max(v):10.3f}")


if __name__ == "__main__":
    main()
