#!/usr/bin/env python3
"""gpurun_out/p (what tools/gpu_run_profiles.sh leaves) -> the small summaries committed under profiles/.

    python tools/condense_profiles.py [--src gpurun_out/p] [--round 2]

Writes profiles/rNN_bench_default.json, rNN_traffic_mppi_cfg2_buffer.json, rNN_<workload>.txt (kernel trace + counters) and
rNN_workloads.txt (one line per bench.py run).  Nothing is computed here: the numbers are the profiler's / bench.py's own."""
import argparse
import contextlib
import glob
import io
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def summarize(stats, *pmc):
    args = [sys.executable, os.path.join(HERE, "summarize_prof.py"), stats] + [p for p in pmc if os.path.exists(p)]
    return subprocess.run(args, capture_output=True, text=True, check=True).stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "p"))
    ap.add_argument("--round", type=int, default=2)
    a = ap.parse_args()
    src, tag = a.src, f"r{a.round:02d}"
    out = os.path.join(ROOT, "profiles")
    wrote = []

    def put(name, text):
        with open(os.path.join(out, name), "w") as f:
            f.write(text)
        wrote.append(name)

    if os.path.exists(os.path.join(src, "bench_200.json")):
        shutil.copy(os.path.join(src, "bench_200.json"), os.path.join(out, f"{tag}_bench_default.json")); wrote.append(f"{tag}_bench_default.json")
    tr = os.path.join(src, f"{tag}_traffic_mppi_cfg2_buffer.json")
    if os.path.exists(tr):
        shutil.copy(tr, os.path.join(out, os.path.basename(tr))); wrote.append(os.path.basename(tr))

    def stats(wl):
        return os.path.join(src, f"prof_{wl}", "p_kernel_stats.csv")

    def pmc(name):
        return os.path.join(src, name, "p_counter_collection.csv")

    plans = {
        "mppi_cfg2": [pmc("pmc_f"), pmc("pmc_w")],
        "rpgd_cfg4": [pmc("pmc_mfma_rpgd_cfg4")],
        "mppi_cfg5_shard": [pmc("pmc_mfma_mppi_cfg5_shard"), pmc("pmc_mfma_mppi_cfg5")],
        "cem_cfg3": [],
        "mppi_gru": [],
    }
    for wl, extra in plans.items():
        if os.path.exists(stats(wl)):
            put(f"{tag}_{wl}.txt", summarize(stats(wl), *extra))

    rows = []
    for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        r = d.get("roofline") or {}
        name = os.path.basename(f)[len("bench_"):-len(".json")]
        rows.append((name, d["ms_per_step"] * 1e3, d.get("step_ms_median", float("nan")) * 1e3, d["value"], r.get("kernel", ""), r.get("kernel_us", float("nan")),
                     r.get("achieved", float("nan")), r.get("unit", ""), r.get("frac", float("nan")), (d.get("config") or {}).get("parallelism", "")))
    if rows:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            print(f"# bench.py lines of round {a.round} (MI355X, gpurun; tools/gpu_run_profiles.sh).  value = trajectory-steps/s at the boundary bench.py names")
            print("# (controller_mpc.step on one GPU; Sharded*.step for the g2 rehearsals: two ranks sharing ONE GPU over gloo — plumbing, not scaling).")
            print(f"{'run':22s} {'us/step':>9s} {'median':>8s} {'value':>10s} {'kernel':46s} {'kern us':>8s} {'achieved':>10s} {'':>7s} {'frac':>7s}  parallelism")
            for n, us, med, val, k, kus, ach, unit, frac, par in rows:
                print(f"{n:22s} {us:9.1f} {med:8.1f} {val:10.3e} {k[:46]:46s} {kus:8.1f} {ach:10.2f} {unit:>7s} {frac:7.4f}  {par[:70]}")
        put(f"{tag}_workloads.txt", buf.getvalue())
    print("wrote:", ", ".join(wrote))


if __name__ == "__main__":
    main()
