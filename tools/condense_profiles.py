#!/usr/bin/env python3
"""What tools/gpu_run_profiles.sh leaves under gpurun_out/p3 -> the small summaries committed under profiles/.

    python3 tools/condense_profiles.py --src gpurun_out/p3 --round 3 --out gpurun_out/p3/condensed     (ON THE GPU BOX, by the script)
    cp gpurun_out/p3/condensed/* profiles/                                                              (here, afterwards)

Nothing is computed here: the numbers are the profiler's / bench.py's own.  Every input must be at least as new as the run's start
stamp (`<src>/run_started`, written by the script before its first step); an older file is REFUSED and named in the output — a summary
can only describe the run that produced its inputs.  (Round 2 condensed a counter CSV of an older kernel: the new one had been deleted on
the box for its size, and an earlier local copy of the same path was picked up.)"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "p3"))
    ap.add_argument("--round", type=int, default=3)
    ap.add_argument("--out", default=None, help="default: <src>/condensed")
    a = ap.parse_args()
    src, tag = a.src, f"r{a.round:02d}"
    out = a.out or os.path.join(src, "condensed")
    os.makedirs(out, exist_ok=True)
    stamp_file = os.path.join(src, "run_started")
    if not os.path.exists(stamp_file):
        raise SystemExit(f"{stamp_file} missing: not the output of tools/gpu_run_profiles.sh")
    started = int(open(stamp_file).read().strip())
    run_id = f"run started {started} (unix s); {open(os.path.join(src, 'summary.txt')).readline().strip() if os.path.exists(os.path.join(src, 'summary.txt')) else ''}"
    wrote, refused = [], []

    def fresh(path):
        """exists and was written by this run"""
        if not os.path.exists(path):
            return False
        if os.path.getmtime(path) + 1 < started:
            refused.append(os.path.relpath(path, src))
            return False
        return True

    def put(name, text):
        with open(os.path.join(out, name), "w") as f:
            f.write(f"# {run_id}\n" + text)
        wrote.append(name)

    def summarize(stats, *pmc):
        args = [sys.executable, os.path.join(HERE, "summarize_prof.py"), stats] + [p for p in pmc if fresh(p)]
        return subprocess.run(args, capture_output=True, text=True, check=True).stdout

    def copy_text(src_name, dst_name, header=""):
        p = os.path.join(src, src_name)
        if fresh(p):
            put(dst_name, header + open(p).read())

    if fresh(os.path.join(src, "bench_200.json")):
        shutil.copy(os.path.join(src, "bench_200.json"), os.path.join(out, f"{tag}_bench_default.json")); wrote.append(f"{tag}_bench_default.json")
    tr = os.path.join(src, f"{tag}_traffic_mppi_cfg2_buffer.json")
    if fresh(tr):
        shutil.copy(tr, os.path.join(out, os.path.basename(tr))); wrote.append(os.path.basename(tr))

    stats = lambda wl: os.path.join(src, f"prof_{wl}", "p_kernel_stats.csv")
    pmc = lambda name: os.path.join(src, name, "p_counter_collection.csv")
    plans = {
        "mppi_cfg2": [pmc("pmc_f"), pmc("pmc_w")],
        "rpgd_cfg4": [pmc("pmc_mfma_rpgd_cfg4")],
        "mppi_cfg5_shard": [pmc("pmc_mfma_mppi_cfg5_shard"), pmc("pmc_mfma_mppi_cfg5")],
        "cem_cfg3": [],
        "mppi_gru": [],
    }
    for wl, extra in plans.items():
        if fresh(stats(wl)):
            put(f"{tag}_{wl}.txt", summarize(stats(wl), *extra))

    # the co-execution microbenchmark: its own print-out + the per-kernel means of its counter pass
    if fresh(os.path.join(src, "coexec.txt")):
        text = "# tools/diag_mfma_coexec (see its header): does independent vector work hide under v_mfma_f32_16x16x4_f32?\n" + open(os.path.join(src, "coexec.txt")).read()
        p = pmc("pmc_coexec")
        if fresh(p):
            import collections
            import csv
            rows = collections.OrderedDict()
            for r in csv.DictReader(open(p)):
                rows.setdefault((r["Kernel_Name"].replace("void ", "").split("(")[0], r["Grid_Size"]), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
            text += "\n# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES (own pass): per-dispatch means\n"
            text += f"{'kernel':28s} {'grid':>7s} {'n':>3s} {'MFMA_BUSY':>12s} {'COEXEC':>8s} {'BUSY_CU':>12s} {'INSTS_VALU':>11s} {'WAVE_CYCLES':>12s}\n"
            for (k, g), c in rows.items():
                if k.startswith("__amd"):
                    continue
                m = lambda n: (sum(c[n]) / len(c[n]) if c[n] else float("nan"))
                text += (f"{k[:28]:28s} {g:>7s} {max(len(v) for v in c.values()):3d} {m('SQ_VALU_MFMA_BUSY_CYCLES'):12.0f} {m('SQ_VALU_MFMA_COEXEC_CYCLES'):8.0f} "
                         f"{m('SQ_BUSY_CU_CYCLES'):12.0f} {m('SQ_INSTS_VALU'):11.0f} {m('SQ_WAVE_CYCLES'):12.0f}\n")
        put(f"{tag}_mfma_coexec.txt", text)

    copy_text("env_kernels.txt", f"{tag}_env_kernels.txt",
              "# tools/bench_env.py on MI355X: CtkEngine.step with the in-kernel sampler, us per step, median (mean) of 300.\n"
              "# \"CartPole tuned\" = generic_kernels off; \"template\" = generic_kernels on (CartPole) / the second environment (Quad2D: 6 states, 2 inputs).\n")
    copy_text("cem_stamps_cfg3.txt", f"{tag}_cem_stamps_cfg3.txt", "# tools/diag_cem_fused (stamped diagnostic build of ctk_cem_fused): where an outer iteration's time goes\n")
    copy_text("cem_stamps_default.txt", f"{tag}_cem_stamps_default.txt", "# tools/diag_cem_fused 200 40 40 (the reference's default CEM size)\n")
    copy_text("sweep_n.txt", f"{tag}_mppi_sweep_n.txt", "# tools/sweep_n.py\n")
    copy_text("resident.txt", f"{tag}_resident.txt", "# tools/bench_resident.py (CtkEngine.step, launched form against the resident form), then tools/diag_mailbox_vram (round trip of one word:\n"
              "# mailbox in pinned host memory against mailbox in host-written device memory)\n")
    copy_text("placement.txt", f"{tag}_wave_placement.txt", "# tools/diag_wave_placement: where a CU puts the waves of small workgroups (every wave: a dependent chain of 14 fp32 MFMAs per step; XCC / SE / CU / SIMD from s_getreg)\n")
    copy_text("parity_margins.txt", f"{tag}_parity_margins.txt", "# tests/margins.py: written by the reference-golden -m gpu tests of this pass (tools/gpu_run_profiles.sh step 0)\n")
    copy_text("rpgd_forms.txt", f"{tag}_rpgd_forms.txt", "# tools/rpgd_forms.py: RPGD + MLP on CartPole's own kernels per MPC step (host clock, 60 steps) — the one-launch form against the phase launches it replaces\n")
    copy_text("rpgd_pers_stamps.txt", f"{tag}_rpgd_pers_stamps.txt", "# tools/rpgd_stamps.sh: wall-clock stamps inside ctk_rpgd_mlp_persistent (variant build; the printf calls perturb what they time: read the lines that agree)\n")
    copy_text("soak_handoff.txt", f"{tag}_soak_handoff.txt", "# tools/soak_handoff.py: the in-launch hand-offs (template wide RPGD descent, the one-launch descent of CartPole's own kernels, block records of MPPI / CEM, the resident kernel) while a second process keeps the GPU busy (3000 steps per environment)\n")
    copy_text("soak.txt", f"{tag}_soak.txt", "# tools/soak.py: closed-loop soak of 29 engines (every optimizer, predictor, environment, the one-launch CEM, the split network kernels, the narrow-record merge, 64-unit and embedded networks, a user environment, the one-launch RPGD forms, the ten-input GRU)\n")
    p = pmc("pmc_largen")
    if fresh(p):
        import collections
        import csv
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            acc[r["Kernel_Name"].replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        N, H = 1 << 20, 50
        text = ("# rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -- python3 tools/large_n_once.py  (MPPI N = 2^20, H = 50, period 1, ODE, sample buffer)\n"
                "# VALU instructions per trajectory-step = SQ_INSTS_VALU (wave instructions) / (N / 64 waves * H steps); whole kernel incl. prologue / epilogue\n"
                f"{'kernel':44s} {'n':>3s} {'SQ_INSTS_VALU':>14s} {'SQ_WAVES':>9s} {'per traj-step':>14s}\n")
        for kname, c in acc.items():
            if kname.startswith("__amd") or kname.startswith("at::"):
                continue
            iv = sum(c["SQ_INSTS_VALU"]) / max(1, len(c["SQ_INSTS_VALU"]))
            wv = sum(c["SQ_WAVES"]) / max(1, len(c["SQ_WAVES"]))
            per = iv / (N / 64 * H) if "rollout" in kname else float("nan")
            text += f"{kname[:44]:44s} {len(c['SQ_INSTS_VALU']):3d} {iv:14.0f} {wv:9.0f} {per:14.1f}\n"
        text += ("# static count of the 16-step loop body (llvm-objdump of ctk_mppi.o): round 2: 1165 VALU of which 135 v_pk_* (half rate) = 72.8 instructions /\n"
                 "# 81 issue slots per trajectory-step; round 3: 1026, none packed = 64.1\n")
        put(f"{tag}_mppi_largeN_insts.txt", text)

    rows = []
    for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        if not fresh(f):
            continue
        try:
            d = json.load(open(f))
        except Exception:
            continue
        r = d.get("roofline") or {}
        name = os.path.basename(f)[len("bench_"):-len(".json")]
        dec = d.get("step_decomposition_us") or {}
        rows.append((name, d["ms_per_step"] * 1e3, d.get("step_ms_median", float("nan")) * 1e3, d["value"], r.get("kernel", ""), r.get("kernel_us", float("nan")),
                     r.get("achieved") or float("nan"), r.get("unit", ""), r.get("frac") or float("nan"), (d.get("config") or {}).get("parallelism", ""),
                     dec.get("begin_kernels"), dec.get("exchange"), dec.get("end_kernels")))
        if name.startswith("g2_") or name.startswith("rccl1") or name.startswith("bare_"):
            shutil.copy(f, os.path.join(out, f"{tag}_bench_{name}.json")); wrote.append(f"{tag}_bench_{name}.json")
    if rows:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            print(f"# bench.py lines of round {a.round} (MI355X, gpurun; tools/gpu_run_profiles.sh).  value = trajectory-steps/s at the boundary bench.py names")
            print("# (controller_mpc.step on one GPU; Sharded*.step for the g2 rehearsals: two ranks sharing ONE GPU over gloo — plumbing, not scaling).")
            print("# begin / exch / end: the sharded step's device-timeline decomposition (stream events; us) — begin kernels | collective | end kernels.")
            print(f"{'run':28s} {'us/step':>9s} {'median':>8s} {'value':>10s} {'kernel':40s} {'kern us':>8s} {'achieved':>10s} {'':>7s} {'frac':>7s} {'begin':>7s} {'exch':>7s} {'end':>7s}  parallelism")
            f1 = lambda v: f"{v:7.1f}" if isinstance(v, (int, float)) else f"{'':>7s}"
            for n, us, med, val, k, kus, ach, unit, frac, par, b, x, e in rows:
                print(f"{n:28s} {us:9.1f} {med:8.1f} {val:10.3e} {k[:40]:40s} {kus:8.1f} {ach:10.2f} {unit:>7s} {frac:7.4f} {f1(b)} {f1(x)} {f1(e)}  {par[:60]}")
        put(f"{tag}_workloads.txt", buf.getvalue())
    if refused:
        put(f"{tag}_REFUSED_STALE_INPUTS.txt", "\n".join(refused) + "\n")
    print("wrote:", ", ".join(wrote))
    if refused:
        print("REFUSED (older than this run):", ", ".join(refused))
        sys.exit(3)


if __name__ == "__main__":
    main()
