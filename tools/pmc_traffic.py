#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py` into the record bench.py reads for `roofline.traffic`:

    profiles/rNN_traffic_<workload>_<samples>.json = {workload, samples, kernel, commit, source_sha256,
                                                      fetch_size_kib, write_size_kib, traffic_bytes_per_launch, ...}

`source_sha256` is bench.kernel_source_digest() of the tree the passes ran on (the GPU box has no .git); bench.py
reports the traffic only while that digest still matches the sources it runs from.

usage (on the GPU box, after the two passes):
  python3 tools/pmc_traffic.py --fetch <..counter_collection.csv> --write <..counter_collection.csv> \
          --workload mppi_cfg2 --samples buffer --commit <sha> --out gpurun_out/r02_traffic_mppi_cfg2_buffer.json"""
import argparse
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mean_counter(path, counter, kernel_prefix):
    vals = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "")
        if r["Counter_Name"] == counter and name.startswith(kernel_prefix):
            vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"{path}: no {counter} rows for a kernel starting with {kernel_prefix!r}")
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True); ap.add_argument("--write", required=True)
    ap.add_argument("--workload", required=True); ap.add_argument("--samples", required=True)
    ap.add_argument("--kernel", default="ctk_mppi_rollout<0, 0, false, false>")
    ap.add_argument("--commit", default="unknown"); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import bench
    prefix = a.kernel.split("(")[0].rstrip(">")     # rocprofv3 prints every template argument: "ctk_mppi_rollout<0, 0, false, false>"
    f, nf = mean_counter(a.fetch, "FETCH_SIZE", prefix)
    w, nw = mean_counter(a.write, "WRITE_SIZE", prefix)
    rec = {"workload": a.workload, "samples": a.samples, "kernel": a.kernel, "commit": a.commit,
           "source_sha256": bench.kernel_source_digest(), "fetch_size_kib": round(f, 3), "write_size_kib": round(w, 3),
           "dispatches": [nf, nw],
           "traffic_bytes_per_launch": int(round((2.0 * f + w) * 1024)),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --no-cpu-baseline "
                     "--no-large-n --no-modes`; KiB per dispatch averaged over the kernel's dispatches; FETCH_SIZE x 2 (gfx950 "
                     "correction of MI355X_MICROARCH.md, confirmed on known-size copies in profiles/r01_fetch_vs_n.txt)"}
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
