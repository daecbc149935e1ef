#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): trajectory-steps/sec = N*H / wall(controller.step).  One "step" = one call of
`controller_mpc.step(s)` (reference Controllers/controller_mpc.py:99-106) -> `optimizer.step` -> ONE pass of the hot
path over one batch of synthetic input, closed loop against a host plant step, logging off.

  python bench.py --gpus 1 --steps 200 --warmup 20            # BASELINE configs[1]: MPPI N=1024, H=50, cart-pole ODE
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus G --steps K --warmup W   # BASELINE configs[4]: MPPI N=65536 (global), H=100, MLP

What the ONE JSON line holds:
  value / ms_per_step   the timed region: K calls at the boundary named in config.boundary, samples as in config.samples
  modes                 the same K-step protocol repeated (outside the timed region) for the other three combinations of
                        boundary in {controller_mpc.step, engine (ctypes ctk_step)} x samples in {buffer, device-rng}:
                        "buffer" = [N,P,C] N(0,1) draws resident in HBM before the region starts (a pool of 16);
                        "device-rng" = drawn inside the step by the in-kernel Philox sampler, as the reference samples
                        inside step() (optimizer_mppi.py:173-175)
  roofline              dominant kernel, timed by its dispatch timestamps in a SEPARATE pass after the timed region
  cpu_baseline          the CPU restatement on this box's host cores (bounded sample): oracle/ctk_cpu.c (C + OpenMP) where it covers the
                        workload, with the single-thread NumPy oracle beside it (cpu_baseline_numpy); else the NumPy oracle

G > 1: strong scaling of BASELINE configs[4] — N = 65536 rollouts split over the ranks, one RCCL all-gather of the
(2+P)-float soft-min record per step (control_toolkit_amd/dist.py).  Other BASELINE configs are parity-test cases;
`--workload` times them too (not the headline line).
"""
import argparse
import glob
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA = fp32 vector rate (spec)

RPGD_KW = dict(outer_its=20, resamp_per=10, shift_previous=1, sampling_distribution=0, sample_min=-1.0, sample_max=1.0,
               learning_rate=0.05, gradmax_clip=5.0)
WORKLOADS = {
    # name: optimizer, predictor, N (GLOBAL rollouts), H, interpolation period, engine kwargs
    "mppi_cfg2": dict(opt="mppi", pred="ODE", N=1024, H=50, p=1, kw={}),
    "mppi_cfg2_interp": dict(opt="mppi", pred="ODE", N=1024, H=50, p=10, kw={}),
    "cem_cfg3": dict(opt="cem", pred="ODE", N=4096, H=30, p=1,
                     kw=dict(cem_outer_it=3, cem_best_k=409, cem_initial_action_stdev=0.5, cem_stdev_min=0.01)),
    "rpgd_cfg4": dict(opt="rpgd", pred="MLP", N=256, H=50, p=10, kw=dict(RPGD_KW, opt_keep_k=64)),
    # BASELINE configs[4]: the sharded configuration (strong scaling: N is split over the ranks)
    "mppi_cfg5": dict(opt="mppi", pred="MLP", N=65536, H=100, p=10, kw={}),
    "mppi_cfg5_shard": dict(opt="mppi", pred="MLP", N=8192, H=100, p=10, kw={}),   # one rank's share at G = 8
    "mppi_cfg5_shard_g4": dict(opt="mppi", pred="MLP", N=16384, H=100, p=10, kw={}),   # ... at G = 4 (rehearsal of the curve's points on one GPU)
    "mppi_cfg5_shard_g2": dict(opt="mppi", pred="MLP", N=32768, H=100, p=10, kw={}),   # ... at G = 2
    # recurrent predictor (2x32 GRU) at the headline MPPI size
    "mppi_gru": dict(opt="mppi", pred="GRU", N=1024, H=50, p=1, kw={}),
    "mppi_mlp": dict(opt="mppi", pred="MLP", N=1024, H=50, p=1, kw={}),
    # the reference's own default problem sizes (Control_Toolkit_ASF_Template/config_optimizers.yml)
    "mppi_default": dict(opt="mppi", pred="ODE", N=3500, H=35, p=10, kw={}),
    "cem_default": dict(opt="cem", pred="ODE", N=200, H=40, p=1,
                        kw=dict(cem_outer_it=3, cem_best_k=40, cem_initial_action_stdev=0.5, cem_stdev_min=0.01)),
    "rpgd_default": dict(opt="rpgd", pred="ODE", N=32, H=40, p=10, kw=dict(RPGD_KW, outer_its=2, opt_keep_k=8)),
    "random_default": dict(opt="random_action", pred="ODE", N=320, H=35, p=1, kw={}),
}

_G, _MC, _MP, _L, _UMAX, _MF, _JF = 9.81, 0.230, 0.087, 0.1975, 2.62, 4.77, 2.5e-4
_INV_MT = 1.0 / (_MC + _MP)
_KML, _KJF, _K43L, _KMM = _MP * _L, _JF / (_MP * _L), _L * (4.0 / 3.0), _MP * _L * _INV_MT


def plant_step(s, u, dt=0.02):
    """Host plant (double-precision cart-pole Euler step, default parameters) that closes the loop so
    the state changes every call.  Bench plumbing only; not the oracle.  Updates `s` in place."""
    x, v, th, om = float(s[0]), float(s[1]), float(s[2]), float(s[3])
    sn, cs = math.sin(th), math.cos(th)
    tmp = (_UMAX * float(u) + _KML * om * om * sn - _MF * v) * _INV_MT
    thdd = (_G * sn - cs * tmp - _KJF * om) / (_K43L - _KMM * cs * cs)
    xdd = tmp - _KMM * thdd * cs
    s[0] = x + dt * v; s[1] = v + dt * xdd; s[2] = th + dt * om; s[3] = om + dt * thdd
    return s


def mlp_weights(seed=0):
    """5-32-32-4 tanh MLP, N(0, 1/fan_in) weights; same recipe as the oracle's."""
    rng = np.random.default_rng(seed)
    parts = [rng.normal(0, 1 / math.sqrt(5), (32, 5)), rng.normal(0, 0.1, (32,)), rng.normal(0, 1 / math.sqrt(32), (32, 32)),
             rng.normal(0, 0.1, (32,)), rng.normal(0, 1 / math.sqrt(32), (4, 32)), rng.normal(0, 0.1, (4,))]
    return np.concatenate([a.ravel() for a in parts]).astype(np.float32)


def gru_weights(seed=0):
    """2x32 GRU + dense 32->4 (10212 floats); same recipe as the oracle's gru_default_weights."""
    rng = np.random.default_rng(seed)
    parts = []
    for fan_in in (5, 32):
        parts += [rng.normal(0, 1 / math.sqrt(fan_in), (96, fan_in)), rng.normal(0, 1 / math.sqrt(32), (96, 32)),
                  rng.normal(0, 0.1, (96,)), rng.normal(0, 0.1, (96,))]
    parts += [rng.normal(0, 1 / math.sqrt(32), (4, 32)), rng.normal(0, 0.1, (4,))]
    return np.concatenate([a.ravel() for a in parts]).astype(np.float32)


def algorithmic(w, N, P, samples_in_hbm, kernel=""):
    """compulsory bytes (and network flops) of ONE launch of the dominant kernel over N rollouts."""
    H, C, S = w["H"], 1, 4
    flops = None
    if w["opt"] == "mppi":
        b = (4 * N * P * C if samples_in_hbm else 0) + 4 * N + 8 * H * C + 4 * S
        if w["pred"] == "MLP":
            flops = 2624 * N * H
        elif w["pred"] == "GRU":   # 2 * (96*5 + 96*32 + 2*96*32 + 4*32) multiply-adds per trajectory step
            flops = 19648 * N * H
    elif w["opt"] == "cem":      # one outer iteration = one rollout launch; the one-launch step (ctk_cem_fused) runs all of them
        K = w["kw"]["cem_best_k"]
        b = (4 * N * H * C if samples_in_hbm else 0) + 4 * N + 4 * K * H * C + 8 * H * C
        if kernel.startswith("ctk_cem_fused"):
            b *= w["kw"]["cem_outer_it"]
    elif w["opt"] == "rpgd":     # one descent launch = outer_its Adam iterations + the final cost pass
        its = w["kw"]["outer_its"]
        b = its * (24 * N * H * C + 4 * N) + 4 * N * H * C + 4 * N
        if w["pred"] == "MLP":
            flops = 2624 * N * H * (2 * its + 1)
    else:
        b = (4 * N * H * C if samples_in_hbm else 0) + 4 * N
    return b, flops


# ------------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle = checker, timed here as the reported CPU port; never part of the product path)
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(w, budget_s=12.0):
    """The oracle (NumPy fp32 restatement of the reference's batched-tensor path) timed on the host
    cores of this box, on a bounded sample of the same workload."""
    from oracle import ctk_oracle as O
    N, H, p = w["N"], w["H"], w["p"]
    pred = O.Predictor(w["pred"])
    cost = O.Cost(pred.env)
    rng = np.random.default_rng(0)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    if w["opt"] == "mppi":
        o = O.MPPI(pred, cost, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
        noise = rng.standard_normal((N, o.P, 1)).astype(np.float32)
        step = lambda s: o.step(s, noise)
    elif w["opt"] == "cem":
        kw = w["kw"]
        o = O.CEM(pred, cost, num_rollouts=N, mpc_horizon=H, cem_outer_it=kw["cem_outer_it"], cem_best_k=kw["cem_best_k"])
        noise = rng.standard_normal((kw["cem_outer_it"], N, H, 1)).astype(np.float32)
        step = lambda s: o.step(s, noise)
    elif w["opt"] == "random_action":
        o = O.RandomAction(pred, cost, num_rollouts=N, mpc_horizon=H)
        u01 = rng.random((N, H, 1), dtype=np.float32)
        step = lambda s: o.step(s, u01)
    else:
        kw = w["kw"]
        o = O.RPGD(pred, cost, num_rollouts=N, mpc_horizon=H, outer_its=kw["outer_its"], resamp_per=kw["resamp_per"],
                   period_interpolation_inducing_points=p, opt_keep_k_ratio=kw["opt_keep_k"] / N)
        o.optimizer_reset(rng.random((N, o.P, 1), dtype=np.float32))
        dr = rng.random((N - o.k, o.P, 1), dtype=np.float32)
        step = lambda s: o.step(s, dr)
    try:                                   # "cores": 1 must hold for the BLAS calls inside NumPy too
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:                      # noqa: BLE001 — not installed: NumPy's own default applies
        limiter = None
    step(s)   # warm-up
    t0 = time.perf_counter(); n = 0
    while True:
        u = step(s); s = plant_step(s.copy(), u); n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    if limiter is not None:
        limiter.restore_original_limits()
    return {"value": N * H * n / el, "unit": "trajectory-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} {w['opt'].upper()} steps of N={N}, H={H} (oracle/ctk_oracle.py, NumPy fp32, single thread), {el:.1f} s"}


def cpu_baseline_native(w, budget_s=6.0):
    """The native multi-core leg (SURVEY 8d iii): oracle/ctk_cpu.c, the C + OpenMP restatement of the same step (held against the
    reference-recorded goldens by tests/test_oracle_cpu_port.py), on this box's host cores.  MPPI / random-action with the CartPole
    analytic predictor only; None otherwise."""
    if w["pred"] != "ODE" or w["opt"] not in ("mppi", "random_action"):
        return None
    from oracle import ctk_oracle as O
    from oracle import ctk_cpu
    N, H, p = w["N"], w["H"], w["p"]
    threads = max(1, min(16, len(os.sched_getaffinity(0)), ctk_cpu.max_threads()))   # a one-GPU box's CPU share is 16 cores
    rng = np.random.default_rng(0)
    env = O.EnvParams()

    def timed(th, budget):
        s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
        if w["opt"] == "mppi":
            c = ctk_cpu.MppiCpu(env, 0.02, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p, threads=th)
            noise = rng.standard_normal((N, c.P)).astype(np.float32)
            step = lambda s: c.step(s, noise)[0]
        else:
            u01 = rng.random((N, H), dtype=np.float32)
            step = lambda s: ctk_cpu.random_action_step(env, 0.02, -1.0, 1.0, s, 0.0, u01, threads=th)[0]
        for _ in range(3):
            step(s)
        t0 = time.perf_counter(); n = 0
        while True:
            u = step(s); s = plant_step(s.copy(), u); n += 1
            el = time.perf_counter() - t0
            if el > budget:
                return N * H * n / el, n, el

    v1, n1, e1 = timed(1, budget_s / 3)
    vt, nt, et = timed(threads, budget_s * 2 / 3)
    best_v, best_t = (vt, threads) if vt >= v1 else (v1, 1)
    return {"value": best_v, "unit": "trajectory-steps/s", "cores": best_t, "kind": "port",
            "sample": f"{nt} {w['opt'].upper()} steps of N={N}, H={H} on {threads} OpenMP threads in {et:.1f} s ({vt:.3e}/s) and {n1} steps on 1 thread in "
                      f"{e1:.1f} s ({v1:.3e}/s) — oracle/ctk_cpu.c (C, gcc -O3, no fast-math, no FMA contraction); the faster of the two is `value`"}


def cpu_baseline_torch(w, budget_s=4.0):
    """Second CPU leg (SURVEY 8d ii): the same MPPI step as batched torch-CPU ops on all host cores — the shape of
    the reference's own PyTorch backend run on a CPU.  At this problem size it is slower than the single-thread
    NumPy port above (a few microseconds of framework overhead on each of the ~2000 small tensor ops of a step)."""
    from oracle import ctk_oracle as O
    from oracle.ctk_oracle_torch import TorchMPPI
    pred = O.Predictor("ODE")
    o = O.MPPI(pred, O.Cost(pred.env), num_rollouts=w["N"], mpc_horizon=w["H"], period_interpolation_inducing_points=w["p"])
    t = TorchMPPI(o, threads=max(1, min(16, len(os.sched_getaffinity(0)))))   # a one-GPU box's CPU share is 16 cores
    noise = np.random.default_rng(0).standard_normal((w["N"], o.P, 1)).astype(np.float32)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    t.step(s, noise)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < budget_s:
        plant_step(s, t.step(s, noise)); n += 1
    el = time.perf_counter() - t0
    return {"value": w["N"] * w["H"] * n / el, "unit": "trajectory-steps/s", "cores": t.threads, "kind": "port",
            "sample": f"{n} MPPI steps of N={w['N']}, H={w['H']} (oracle/ctk_oracle_torch.py, torch-CPU fp32 batched ops, "
                      f"{t.threads} threads), {el:.1f} s"}


# ------------------------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel: read from the newest profiles/*traffic*.json that was taken from THESE sources
# ------------------------------------------------------------------------------------------------------------------
# kernel sources AND what selects the kernel / sizes its buffers (fuse mode, thresholds, ABI structs)
KERNEL_SOURCES = ("ctk_mppi.hip", "ctk_mppi_body.inc", "ctk_mppi_body_1_decl.inc", "ctk_mppi_body_2_pro1.inc", "ctk_mppi_body_3_defs.inc",
                  "ctk_mppi_body_4_phase_a.inc", "ctk_mppi_body_5_post.inc", "ctk_mppi_merge.h", "ctk_env.h", "ctk_rollout.h", "ctk_device.h", "ctk_common.h", "ctk_mlp.h", "ctk_gru.h", "Makefile",
                  "ctk_api.hip", "ctk_launch.h", "../../include/ctk_hip.h")


def kernel_source_digest():
    """sha256 over the sources the MPPI rollout kernel is compiled from (the GPU box has no .git: a content hash, not a
    commit id, is what can be checked there)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        path = os.path.join(ROOT, "control_toolkit_amd", "csrc", name)
        if os.path.exists(path):
            h.update(name.encode()); h.update(open(path, "rb").read())
    return h.hexdigest()


def recorded_traffic(workload, samples, kernel):
    """(bytes per launch or None, note).  A record is used only if it names this workload / sample mode / kernel AND the
    kernel sources have not changed since it was taken (tools/pmc_traffic.py writes the digest) — otherwise null."""
    digest = kernel_source_digest()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True)
    stale = None
    for path in cands:
        try:
            rec = json.load(open(path))
        except Exception:   # noqa: BLE001
            continue
        if rec.get("workload") != workload or rec.get("samples") != samples or rec.get("kernel") != kernel:
            continue
        if rec.get("source_sha256") != digest:
            stale = stale or os.path.basename(path)
            continue
        return rec.get("traffic_bytes_per_launch"), (f"{os.path.basename(path)} (commit {rec.get('commit', '?')}, kernel sources unchanged since): "
                                                      f"2 x FETCH_SIZE {rec.get('fetch_size_kib')} KiB + WRITE_SIZE {rec.get('write_size_kib')} KiB per launch, "
                                                      "separate rocprofv3 --pmc passes, the guide's gfx950 factor 2 on FETCH_SIZE")
    if stale:
        return None, f"no valid record: {stale} was taken from older kernel sources (re-run tools/pmc_traffic.py)"
    return None, "no PMC record for this workload / sample mode under profiles/"


# ------------------------------------------------------------------------------------------------------------------
def large_n_point(torch, CtkEngine, dev, H, p, N=1 << 20, steps=12):
    """The same MPPI step at N = 2^20 (outside the timed region, not part of `value`): where the path sits against
    the HBM roofline once the chip is full.  BASELINE's size occupies 16 of 256 CUs, so its own fraction says
    nothing about the kernel's efficiency; this does (scaled-N sweep)."""
    eng = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=1)
    P = eng.mppi_partial_size() - 2
    buf = torch.randn((N, P, 1), device=dev, dtype=torch.float32)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for _ in range(3):
        eng.step(s, buf.data_ptr())
    eng.profile_enable(True, every=1)
    for _ in range(steps):
        eng.step(s, buf.data_ptr())
    k_ms = float(np.mean(eng.profile_read()))
    name = eng.dominant_kernel()
    eng.close()
    alg = 4 * N * P + 4 * N + 8 * H + 16
    ach = alg / (k_ms * 1e-3) / 1e9
    return {"N": N, "kernel": name, "kernel_us": k_ms * 1e3, "bound": "valu-issue", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "trajectory_steps_per_s_kernel": N * H / (k_ms * 1e-3),
            "note": "issue/latency-bound before it is HBM-bound (DESIGN.md 5: ~75 instructions per trajectory-step vs 4 compulsory bytes)"}


def build_controller(w, N_local, local_rank, rank, rng_mode):
    """controller_mpc around `<optimizer>-hip` exactly as a reference caller would build it
    (Controllers/controller_mpc.py:24-96; tests/test_gpu_controller.py does the same against the golden closed loop)."""
    from control_toolkit_amd.Controllers.controller_mpc import controller_mpc
    from control_toolkit_amd.Predictors import PredictorWrapper
    from control_toolkit_amd.Cost_Functions import CostFunctionWrapper
    name = {"mppi": "mppi-hip", "cem": "cem-hip", "rpgd": "rpgd-hip", "random_action": "random-action-hip"}[w["opt"]]
    H, p, kw = w["H"], w["p"], w["kw"]
    common = dict(seed=1, mpc_horizon=H, num_rollouts=N_local, mpc_timestep=0.02, rng_mode=rng_mode, device=local_rank)
    if w["opt"] == "mppi":
        oc = dict(common, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03, period_interpolation_inducing_points=p,
                  global_rollout_offset=rank * N_local)
    elif w["opt"] == "cem":
        oc = dict(common, cem_outer_it=kw["cem_outer_it"], cem_initial_action_stdev=kw["cem_initial_action_stdev"],
                  cem_stdev_min=kw["cem_stdev_min"], cem_best_k=kw["cem_best_k"], warmup=False, warmup_iterations=0)
    elif w["opt"] == "rpgd":
        oc = dict(common, outer_its=kw["outer_its"], sample_stdev=0.5, sample_mean=0.0, sample_whole_control_space=True,
                  uniform_dist_min=-1.0, uniform_dist_max=1.0, resamp_per=kw["resamp_per"], period_interpolation_inducing_points=p,
                  SAMPLING_DISTRIBUTION="uniform", shift_previous=kw["shift_previous"], warmup=False, warmup_iterations=0,
                  learning_rate=kw["learning_rate"], opt_keep_k_ratio=kw["opt_keep_k"] / N_local, gradmax_clip=kw["gradmax_clip"],
                  rtol=1e-3, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-8)
    else:
        oc = dict(common)
    weights = mlp_weights(0) if w["pred"] == "MLP" else gru_weights(0) if w["pred"] == "GRU" else None
    cc = {"mpc": {"optimizer": name, "predictor_specification": w["pred"], "cost_function_specification": "default",
                  "computation_library": "hip", "controller_logging": False, "calculate_optimal_trajectory": False,
                  "device": f"gpu:{local_rank}"}}
    c = controller_mpc("CartPole", (np.array([-1.0], np.float32), np.array([1.0], np.float32)),
                       {"target_position": 0.0, "target_equilibrium": 1.0}, config_controllers=cc,
                       config_optimizers={name: oc}, predictor=PredictorWrapper(weights=weights),
                       cost_function=CostFunctionWrapper(watch=False))
    c.configure()
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: mppi_cfg2 (BASELINE configs[1]) on one GPU, mppi_cfg5 (configs[4], N split over the ranks) on several")
    ap.add_argument("--samples", default="buffer", choices=["buffer", "device-rng"],
                    help="sample mode of the TIMED region (both are reported): buffers resident in HBM, or the in-kernel Philox sampler")
    ap.add_argument("--boundary", default="controller", choices=["controller", "engine"],
                    help="boundary of the TIMED region (both are reported): controller_mpc.step or the ctypes engine")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-n", action="store_true", help="skip the scaled-N roofline point (same kernel family, N = 2^20)")
    ap.add_argument("--no-modes", action="store_true", help="skip the three extra boundary x samples passes")
    ap.add_argument("--force-sharded", action="store_true", help="use the begin / all-gather / end path even with one rank")
    args = ap.parse_args()

    # `python bench.py --gpus G` as a bare command (no launcher in the environment): start the one-process-per-GPU job as a CHILD process
    # (never exec: this process may not touch the GPU before, and must not be replaced after), relay rank 0's JSON line and the child's
    # return code.  Nothing above has initialised the GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
        line = None
        for ln in child.stdout.decode(errors="replace").splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)
        if line is not None:
            print(line, flush=True)
        raise SystemExit(child.returncode if child.returncode else (0 if line is not None else 1))

    # Contract: ONE JSON line on stdout.  Native libraries write banners to fd 1 (RCCL prints its version block
    # there at communicator init), so everything but the result line goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from control_toolkit_amd import CtkEngine
    from control_toolkit_amd.dist import ShardedMPPI, ShardedTopK, ShardedRPGD
    from control_toolkit_amd.others.globals_and_utils import DeviceBufferRng, DeviceRng

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (torch.distributed.run --nproc-per-node {args.gpus}), or run "
                         f"`python bench.py --gpus {args.gpus}` without a launcher and it starts the ranks itself")
    # rehearsal hooks for a one-GPU box (never set by the driver): all ranks on device 0, gloo instead of RCCL
    backend = os.environ.get("CTK_BENCH_BACKEND", "nccl")
    if os.environ.get("CTK_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # rehearsal hook: a ONE-rank RCCL group with the collective actually issued, to exercise the RCCL call path
    force_pg = os.environ.get("CTK_BENCH_FORCE_PG") == "1"
    use_pg = world > 1 or force_pg
    if use_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wname = args.workload or ("mppi_cfg2" if world == 1 else "mppi_cfg5")
    w = WORKLOADS[wname]
    Ng, H, p = w["N"], w["H"], w["p"]
    if Ng % world:
        raise SystemExit(f"{wname}: {Ng} rollouts do not split evenly over {world} ranks")
    N = Ng // world                                          # strong scaling: this rank's shard
    sharded_run = world > 1 or args.force_sharded or force_pg
    if sharded_run and w["opt"] == "rpgd" and w["kw"]["opt_keep_k"] > Ng:
        raise SystemExit("opt_keep_k exceeds the population")

    # ---- the product objects: controller_mpc (+ its engine) for the single-GPU boundary, Sharded* for G > 1 -------
    ekw = dict(w["kw"])
    ctrl = None
    if not sharded_run:
        ctrl = build_controller(w, N, local_rank, rank, "device")
        eng = ctrl.optimizer.engine
    else:
        eng = CtkEngine(w["opt"], w["pred"], num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                        seed=1, device=local_rank, global_rollout_offset=rank * N, **ekw)
        if w["pred"] == "MLP":
            eng.set_predictor_weights(mlp_weights(0))
        elif w["pred"] == "GRU":
            eng.set_predictor_weights(gru_weights(0))
    P = eng.mppi_partial_size() - 2
    sharded, exchange_note = None, "single GPU"
    if sharded_run:
        if w["opt"] == "mppi":
            want = os.environ.get("CTK_BENCH_EXCHANGE", "rccl")   # headline: the RCCL all-gather north_star names; p2p is opt-in
            sharded = ShardedMPPI(eng, rank, world, device=dev, always_collective=force_pg, exchange=want)
            if want == "p2p" and sharded.exchange != "p2p":
                exchange_note = (f"rollout shards x{world}, 1 all-gather of {P + 2} floats per step (RCCL) — peer-to-peer exchange was "
                                 f"requested but unavailable ({sharded.p2p_error}); fell back collectively")
            elif sharded.exchange == "p2p":
                exchange_note = f"rollout shards x{world}, records of {P + 2} floats exchanged by peer-to-peer stores over xGMI (opt-in)"
            else:
                exchange_note = f"rollout shards x{world}, 1 all-gather of {P + 2} floats per step (RCCL)"
        elif w["opt"] in ("cem", "random_action"):
            sharded = ShardedTopK(eng, rank, world, device=dev)
            exchange_note = f"rollout shards x{world}, 1 all-gather of {sharded.rec} floats (best-K records) per outer iteration (RCCL)"
        else:
            eng.reset()
            sharded = ShardedRPGD(eng, rank, world, device=dev)
            exchange_note = f"rollout shards x{world}, 1 all-gather of {sharded.rec} floats (keeper records) per step (RCCL)"

    # synthetic inputs, resident in HBM before any timed region: a pool of sample buffers
    need = max(int(eng.samples_needed()), N * max(P, H), 1)
    if w["opt"] == "rpgd":
        need = max(need, N * P)
    g = torch.Generator(device=dev); g.manual_seed(1 + rank)
    uniform = w["opt"] in ("random_action", "rpgd")
    pool = [(torch.rand if uniform else torch.randn)((need,), generator=g, device=dev, dtype=torch.float32) for _ in range(16)]
    ptrs = [t.data_ptr() for t in pool]
    rng0 = np.random.default_rng(0)
    s0 = np.array([rng0.uniform(-0.2, 0.2), rng0.uniform(-0.5, 0.5), rng0.uniform(-np.pi, np.pi), rng0.uniform(-2, 2)], np.float32)

    def make_step(boundary, samples):
        """a callable s -> u for one boundary x sample-mode combination"""
        if sharded is not None:
            if w["opt"] == "rpgd":
                return lambda s, i: sharded.step(s, None)          # fresh rows by device Philox (global row index)
            if samples == "buffer" and w["opt"] == "mppi":
                return lambda s, i: sharded.step(s, ptrs[i & 15])
            return lambda s, i: sharded.step(s, None)
        if boundary == "controller":
            ctrl.optimizer.rng = DeviceBufferRng(ptrs, seed=1) if samples == "buffer" else DeviceRng(1)
            return lambda s, i: ctrl.step(s)
        if w["opt"] == "rpgd":                                     # the engine decides when it needs draws
            return (lambda s, i: eng.step(s, ptrs[i & 15] if eng.samples_needed() else None)) if samples == "buffer" else (lambda s, i: eng.step(s, None))
        return (lambda s, i: eng.step(s, ptrs[i & 15])) if samples == "buffer" else (lambda s, i: eng.step(s, None))

    bracket = {}

    def run_region(step_fn, steps, warmup, timed_region, before_close=None):
        """W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks.
        before_close: called INSIDE the timed region right before the closing synchronize (the resident pass ends its kernel there)."""
        s = s0.copy()
        # a fixed 64-step priming before the W warm-up steps (clock ramp, code / descriptor caches, Python's specialising
        # interpreter): the driver's short runs (W = 5, K = 20) then time the same steady state as a 200-step run
        for i in range(64 + warmup):
            plant_step(s, np.asarray(step_fn(s, i)).reshape(-1)[0])
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        per_step = np.empty(steps)
        t0 = time.perf_counter(); ta = t0
        for i in range(steps):
            plant_step(s, np.asarray(step_fn(s, i)).reshape(-1)[0])   # controller.step, then the plant: closed loop
            tb = time.perf_counter(); per_step[i] = tb - ta; ta = tb
        if before_close is not None:
            before_close()
        t_sync = time.perf_counter()
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        bracket["close_ms"] = (time.perf_counter() - t_sync) * 1e3     # the closing synchronize (+ barrier): inside the bracket by contract
        if use_pg:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, per_step

    boundary = "sharded" if sharded is not None else args.boundary
    samples = args.samples
    elapsed, per_step = run_region(make_step(boundary, samples), args.steps, args.warmup, True)
    bracket_close_ms = bracket.get("close_ms")

    # ---- the resident form (opt-in, include/ctk_hip.h: ctk_resident_*): the same K-step protocol at the same boundary with the steps served
    #      by a kernel that stays on the device.  A SEPARATE field: `value` above stays one launch per step.  The kernel is ended inside the
    #      timed region, before the closing synchronize (which would otherwise wait for its idle time-out).
    resident = None
    if sharded is None and w["opt"] == "mppi" and w["pred"] == "ODE" and not args.no_modes:
        try:
            r_eng = ctrl.optimizer.engine if boundary == "controller" else eng
            # buffers: read when the request arrives (the default, safe for a buffer refilled in place); the pool here is static, so the
            # read-ahead opt-in (ctk_resident_enable(h, 2, ...)) applies as well and is reported beside it
            r_eng.resident_enable(True, 500.0)
            el_r, ps_r = run_region(make_step(boundary, samples), args.steps, min(args.warmup, 5), False, before_close=r_eng.resident_stop)
            st_r = r_eng.resident_stats()
            r_eng.resident_enable(False)
            ahead = None
            if samples == "buffer":
                r_eng.resident_enable(True, 500.0, read_ahead=True)
                el_a, _ = run_region(make_step(boundary, samples), args.steps, min(args.warmup, 5), False, before_close=r_eng.resident_stop)
                r_eng.resident_enable(False)
                ahead = {"value": Ng * H * args.steps / el_a, "ms_per_step": el_a / args.steps * 1e3,
                         "note": "read_ahead=True: the caller promises a static sample pool, so the next step's draws are read between steps"}
            resident = {"value": Ng * H * args.steps / el_r, "ms_per_step": el_r / args.steps * 1e3, "step_ms_median": float(np.median(ps_r) * 1e3),
                        "step_ms_p95": float(np.percentile(ps_r, 95) * 1e3), "boundary": "controller_mpc.step" if boundary == "controller" else "engine.step",
                        "samples": samples, "sample_buffers": "read at the request (no read-ahead)" if samples == "buffer" else "in-kernel sampler",
                        "static_pool_read_ahead": ahead,
                        "idle_us": 500.0, "mailbox": st_r["mailbox"], "kernel_launches_total": st_r["launches"],
                        "note": "opt-in resident kernel fed through a mailbox: same results bit for bit (tests/test_gpu_resident.py); it is ended inside the "
                                "timed region before the closing synchronize; NOT the headline `value`, which stays one launch per step"}
        except Exception as ex:                                           # noqa: BLE001 — the optional pass must never take the line down
            resident = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- the other boundary x sample-mode combinations, same protocol, outside the timed region ------------------
    modes = {}
    if sharded is None and not args.no_modes:
        for b in ("controller", "engine"):
            for sm in ("buffer", "device-rng"):
                if (b, sm) == (boundary, samples):
                    el, ps = elapsed, per_step
                else:
                    el, ps = run_region(make_step(b, sm), args.steps, min(args.warmup, 5), False)
                modes[f"{'controller_mpc.step' if b == 'controller' else 'engine.step'}/{sm}"] = {
                    "value": Ng * H * args.steps / el, "ms_per_step": el / args.steps * 1e3, "step_ms_median": float(np.median(ps) * 1e3)}

    # ---- dominant-kernel time: a separate pass with every launch timed through its dispatch timestamps (timing a
    #      launch costs host time, so it stays out of the timed region) --------------------------------------------
    kstep = make_step("sharded" if sharded is not None else "engine", samples)
    kpass = max(8, min(64, args.steps))
    s = s0.copy()
    eng.profile_enable(True, every=1)
    if sharded is not None:
        sharded.enable_timing(True)          # four stream events per step: begin kernels | exchange | end kernels
    for i in range(kpass):
        plant_step(s, np.asarray(kstep(s, i)).reshape(-1)[0])
    kern_ms = eng.profile_read()
    eng.profile_enable(False)
    exch = None
    if sharded is not None:
        torch.cuda.synchronize()
        exch = sharded.timing_us()
        sharded.enable_timing(False)
    # every rank's dominant-kernel time and exchange time to rank 0 (a scaling curve is read per rank: the slowest one sets the step)
    per_rank = None
    if use_pg:
        mine = torch.tensor([float(np.mean(kern_ms)) * 1e3 if len(kern_ms) else float("nan"),
                             float(np.mean(exch["begin_us"])) if exch and exch["begin_us"] else float("nan"),
                             float(np.mean(exch["exchange_us"])) if exch and exch["exchange_us"] else float("nan"),
                             float(np.mean(exch["end_us"])) if exch and exch["end_us"] else float("nan")], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in t.cpu()] for t in allr]

    if rank == 0:
        total_units = Ng * H * args.steps
        kms = float(np.mean(kern_ms)) if len(kern_ms) else float("nan")
        samples_in_hbm = samples == "buffer" and w["opt"] in ("mppi", "cem", "random_action")
        kname = eng.dominant_kernel()
        alg_bytes, alg_flops = algorithmic(w, N, P, samples_in_hbm, kname)
        ok = kms == kms and kms > 0
        if alg_flops is not None:
            ach = alg_flops / (kms * 1e-3) / 1e12 if ok else None
            roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / MFMA_F32_PEAK_TF if ach else None, "traffic": None, "algorithmic_flops": alg_flops,
                    "note": "fp32-input MFMA (v_mfma_f32_16x16x4_f32) for exact-fp32 parity; per launch of this rank's shard"}
        else:
            ach = alg_bytes / (kms * 1e-3) / 1e9 if ok else None
            traffic, tnote = recorded_traffic(wname, samples, kname)
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": traffic, "traffic_source": tnote,
                    "note": "issue/latency-bound at this size, not HBM-bound: 0.2 MB per launch vs an H-step dependent recurrence on one "
                            "wave per 64 trajectories (DESIGN.md 5); the large-N point below is where the chip is full"}
        roof.update({"kernel": kname, "kernel_us": kms * 1e3, "algorithmic_bytes": alg_bytes,
                     "kernel_timed_launches": int(len(kern_ms)), "kernel_timing": "separate pass after the timed region, every launch"})
        sharded_fields = {}
        if sharded is not None:
            # where a sharded step's time goes, on the device timeline of this rank (events on the stream the engine and the collective share):
            # begin = rollout (+ local merge / selection), exchange = the collective incl. the wait for the slowest peer, end = merge / update
            mean = lambda k: (float(np.mean(exch[k])) if exch and exch[k] else None)
            sharded_fields["exchange_us"] = mean("exchange_us")
            sharded_fields["step_decomposition_us"] = {"begin_kernels": mean("begin_us"), "exchange": mean("exchange_us"), "end_kernels": mean("end_us"),
                                                       "exchange_kind": ("p2p stores" if getattr(sharded, "exchange", "rccl") == "p2p" else
                                                                         ("RCCL all-gather" if backend == "nccl" else f"{backend} all-gather (rehearsal)") if use_pg else "none (one rank)"),
                                                       "note": "stream events around begin / collective / end in the kernel-timing pass; p2p: one launch, not decomposed"}
            if per_rank is not None:
                unit_alg = alg_flops if alg_flops is not None else alg_bytes
                scale = 1e12 if alg_flops is not None else 1e9
                peak = MFMA_F32_PEAK_TF if alg_flops is not None else HBM_PEAK_GBS
                sharded_fields["roofline_per_rank"] = [
                    {"rank": r, "kernel_us": v[0], "achieved": (unit_alg / (v[0] * 1e-6) / scale) if v[0] == v[0] and v[0] > 0 else None,
                     "frac": (unit_alg / (v[0] * 1e-6) / scale / peak) if v[0] == v[0] and v[0] > 0 else None,
                     "begin_us": v[1], "exchange_us": v[2], "end_us": v[3]} for r, v in enumerate(per_rank)]
        ps = per_step * 1e3
        out = {
            "metric": "trajectory-steps/sec (N*H per controller.step)", "value": total_units / elapsed,
            "unit": "trajectory-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming": 64,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w['opt'].upper()} N={Ng} global ({N} per GPU), H={H}, period={p}, predictor {w['pred']} "
                                   f"(4 states, 1 input) [{wname}]",
                       "boundary": {"controller": "controller_mpc.step (Python; s in, u out on the host)",
                                    "engine": "CtkEngine.step (ctypes -> ctk_step)",
                                    "sharded": "control_toolkit_amd.dist.Sharded*.step (ctypes begin / collective / end)"}[boundary],
                       "samples": samples if (sharded is None or w["opt"] == "mppi") else "device-rng",
                       "global_rollouts": Ng, "parallelism": exchange_note},
            "step_ms_median": float(np.median(ps)),
            "roofline": roof,
        }
        out.update(sharded_fields)
        # what the barrier + synchronize bracket itself contributes to `ms_per_step` (fixed per region, so 1/K of it per step: a K = 20
        # run reads ~2 us per step higher than a K = 200 run of the same steady state; the median does not move)
        if os.environ.get("CTK_BENCH_TRACE_STEPS"):
            print("per-step us of the timed region:", " ".join(f"{x * 1e3:.1f}" for x in ps[:24]), file=sys.stderr)
        out["timed_region"] = {"first_step_ms": float(ps[0]), "closing_synchronize_ms": bracket_close_ms,
                               "ms_per_step_without_first_and_close": float((elapsed * 1e3 - ps[0] - (bracket_close_ms or 0.0)) / max(1, args.steps - 1))}
        if args.steps >= 100:
            out["step_ms_p95"] = float(np.percentile(ps, 95))
        if modes:
            out["modes"] = modes
        if resident is not None:
            out["resident"] = resident
        if not args.no_large_n and world == 1 and wname == "mppi_cfg2":
            out["roofline_large_n"] = large_n_point(torch, CtkEngine, dev, H, p)
        if not args.no_cpu_baseline and world == 1:
            numpy_leg = cpu_baseline(w, budget_s=8.0)
            native = cpu_baseline_native(w)
            # `cpu_baseline` = the fastest CPU restatement available for this workload (the native C + OpenMP port where it exists);
            # the single-thread NumPy oracle stays on the line beside it
            out["cpu_baseline"] = native if native is not None else numpy_leg
            if native is not None:
                out["cpu_baseline_numpy"] = numpy_leg
            if w["opt"] == "mppi" and w["pred"] == "ODE":
                out["cpu_baseline_torch"] = cpu_baseline_torch(w)
        if world > 1 and w["opt"] == "mppi":
            # the ONE-GPU point of the same workload, measured on rank 0's GPU while the others wait: lets a reader
            # turn the strong-scaling value into an efficiency without a second launch of this script
            e1 = CtkEngine("mppi", w["pred"], num_rollouts=Ng, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                           seed=1, device=local_rank, **ekw)
            if w["pred"] == "MLP":
                e1.set_predictor_weights(mlp_weights(0))
            elif w["pred"] == "GRU":
                e1.set_predictor_weights(gru_weights(0))
            big = torch.randn((Ng * P,), device=dev, dtype=torch.float32)
            # the SAME boundary as the sharded step (Sharded*.step: begin / [no collective for one rank] / end), so that value / this
            # value is a like-for-like strong-scaling ratio
            one = ShardedMPPI(e1, 0, 1, device=dev)
            s = s0.copy()
            for _ in range(5):
                plant_step(s, one.step(s, big.data_ptr() if samples == "buffer" else None)[0])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n1 = min(args.steps, 50)
            for _ in range(n1):
                plant_step(s, one.step(s, big.data_ptr() if samples == "buffer" else None)[0])
            torch.cuda.synchronize(); el1 = time.perf_counter() - t0
            out["single_gpu_same_workload"] = {"value": Ng * H * n1 / el1, "ms_per_step": el1 / n1 * 1e3, "steps": n1,
                                               "boundary": "control_toolkit_amd.dist.ShardedMPPI.step with one rank (begin / end, no collective)",
                                               "note": "the whole N on rank 0's GPU, outside the timed region, while the other ranks wait"}
            e1.close()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if use_pg:
        dist.barrier()
    if ctrl is None:
        eng.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
