#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): trajectory-steps/sec = N*H / wall(controller.step).  Default workload =
BASELINE configs[1]: MPPI, N=1024, H=50, 4-state analytic cart-pole; one "step" = one full MPPI
iteration (sample buffer resident in HBM -> fused rollout+cost -> soft-min merge/update -> u back on
the host), timed at the optimizer.step boundary, closed loop against a host plant step.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: weak scaling — every rank rolls out its own shard (global population = N_local * ranks), one
all-gather of the (2+P)-float soft-min record per step over RCCL (control_toolkit_amd/dist.py).
Other BASELINE configs are parity-test cases; `--workload` times them too (not the headline line).
"""
import argparse
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

WORKLOADS = {
    # name: dict(optimizer, predictor, N, H, period, engine kwargs, rollouts per step fwd/bwd)
    "mppi_cfg2": dict(opt="mppi", pred="ODE", N=1024, H=50, p=1, kw={}),
    "mppi_cfg2_interp": dict(opt="mppi", pred="ODE", N=1024, H=50, p=10, kw={}),
    "cem_cfg3": dict(opt="cem", pred="ODE", N=4096, H=30, p=1,
                     kw=dict(cem_outer_it=3, cem_best_k=409, cem_initial_action_stdev=0.5, cem_stdev_min=0.01)),
    "rpgd_cfg4": dict(opt="rpgd", pred="MLP", N=256, H=50, p=10,
                      kw=dict(outer_its=20, resamp_per=10, shift_previous=1, opt_keep_k=64, sampling_distribution=0,
                              sample_min=-1.0, sample_max=1.0, learning_rate=0.05, gradmax_clip=5.0)),
    "mppi_cfg5_shard": dict(opt="mppi", pred="MLP", N=8192, H=100, p=10, kw={}),
    # recurrent predictor (2x32 GRU, weights in LDS) at the headline MPPI size
    "mppi_gru": dict(opt="mppi", pred="GRU", N=1024, H=50, p=1, kw={}),
    "mppi_mlp": dict(opt="mppi", pred="MLP", N=1024, H=50, p=1, kw={}),
    # the reference's own default problem sizes (Control_Toolkit_ASF_Template/config_optimizers.yml)
    "mppi_default": dict(opt="mppi", pred="ODE", N=3500, H=35, p=10, kw={}),
    "cem_default": dict(opt="cem", pred="ODE", N=200, H=40, p=1,
                        kw=dict(cem_outer_it=3, cem_best_k=40, cem_initial_action_stdev=0.5, cem_stdev_min=0.01)),
    "rpgd_default": dict(opt="rpgd", pred="ODE", N=32, H=40, p=10,
                         kw=dict(outer_its=2, resamp_per=10, shift_previous=1, opt_keep_k=8, sampling_distribution=0,
                                 sample_min=-1.0, sample_max=1.0, learning_rate=0.05, gradmax_clip=5.0)),
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


def algorithmic(w, P, samples_in_hbm):
    """compulsory bytes (and MLP flops) of the DOMINANT kernel's launch."""
    N, H, C, S = w["N"], w["H"], 1, 4
    flops = None
    if w["opt"] == "mppi":
        b = (4 * N * P * C if samples_in_hbm else 0) + 4 * N + 8 * H * C + 4 * S
        if w["pred"] == "MLP":
            flops = 2624 * N * H
        elif w["pred"] == "GRU":   # 2 * (96*5 + 96*32 + 2*96*32 + 4*32) multiply-adds per trajectory step
            flops = 19648 * N * H
    elif w["opt"] == "cem":      # one outer iteration = one rollout launch
        K = w["kw"]["cem_best_k"]
        b = (4 * N * H * C if samples_in_hbm else 0) + 4 * N + 4 * K * H * C + 8 * H * C
    elif w["opt"] == "rpgd":     # one descent launch = outer_its Adam iterations + the final cost pass
        its = w["kw"]["outer_its"]
        b = its * (24 * N * H * C + 4 * N) + 4 * N * H * C + 4 * N
        if w["pred"] == "MLP":
            flops = 2624 * N * H * (2 * its + 1)
    else:
        b = 4 * N * H * C + 4 * N
    return b, flops


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
    return {"N": N, "kernel": name, "kernel_us": k_ms * 1e3, "bound": "valu", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "trajectory_steps_per_s_kernel": N * H / (k_ms * 1e-3),
            "note": "VALU-bound before it is HBM-bound: ~60 fp32 instructions incl. sin/cos per 4 compulsory bytes; PMC: VALU "
                    "utilisation 76 % at this size (profiles/r01_mppi_largeN_pmc.txt)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="mppi_cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", default="buffer", choices=["buffer", "device-rng"],
                    help="MPPI: [N,P,C] N(0,1) sample buffers resident in HBM (north_star) or the in-kernel Philox sampler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-n", action="store_true", help="skip the scaled-N roofline point (same kernel family, N = 2^20)")
    ap.add_argument("--force-sharded", action="store_true", help="use the begin / all-gather / end path even with one rank")
    args = ap.parse_args()

    # Contract: ONE JSON line on stdout.  Native libraries write banners to fd 1 (RCCL prints its version block
    # there at communicator init), so everything but the result line goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from control_toolkit_amd import CtkEngine
    from control_toolkit_amd.dist import ShardedMPPI

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # rehearsal hooks for a one-GPU box (never set by the driver): all ranks on device 0, gloo instead of RCCL
    backend = os.environ.get("CTK_BENCH_BACKEND", "nccl")
    if os.environ.get("CTK_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # rehearsal hook: a ONE-rank RCCL group with the collective actually issued, to exercise the RCCL call path
    # (init, all_gather_into_tensor / all_reduce / barrier on torch's stream) on a one-GPU box
    force_pg = os.environ.get("CTK_BENCH_FORCE_PG") == "1"
    if world > 1 or force_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    w = WORKLOADS[args.workload]
    if world > 1 and w["opt"] != "mppi":
        raise SystemExit("only the MPPI workloads are sharded (DESIGN.md 6)")
    N, H, p = w["N"], w["H"], w["p"]
    eng = CtkEngine(w["opt"], w["pred"], num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                    seed=1, device=local_rank, global_rollout_offset=rank * N, **w["kw"])
    if w["pred"] == "MLP":
        eng.set_predictor_weights(mlp_weights(0))
    elif w["pred"] == "GRU":
        eng.set_predictor_weights(gru_weights(0))
    P = eng.mppi_partial_size() - 2
    sharded = None
    if world > 1 or args.force_sharded or force_pg:
        # the collective runs on torch's stream: issue the engine's kernels there too
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        # record exchange between the ranks: direct peer-to-peer stores over xGMI (falls back to the RCCL all-gather,
        # collectively, if the IPC set-up or its self-test fails on any rank); CTK_BENCH_EXCHANGE=rccl forces RCCL
        sharded = ShardedMPPI(eng, rank, world, device=dev, always_collective=force_pg,
                              exchange=os.environ.get("CTK_BENCH_EXCHANGE", "p2p"))
        if rank == 0 and sharded.p2p_error:
            print(f"[bench] p2p exchange unavailable ({sharded.p2p_error}); using the RCCL all-gather", file=sys.stderr)
    if w["opt"] == "rpgd":
        eng.reset()

    # synthetic inputs, resident in HBM before the timed region: a pool of sample buffers (MPPI)
    pool = None
    samples_in_hbm = w["opt"] == "mppi" and args.samples == "buffer"
    if samples_in_hbm:
        g = torch.Generator(device=dev); g.manual_seed(1 + rank)
        pool = [torch.randn((N, P, 1), generator=g, device=dev, dtype=torch.float32) for _ in range(16)]
        ptrs = [t.data_ptr() for t in pool]
    rng0 = np.random.default_rng(0)
    s = np.array([rng0.uniform(-0.2, 0.2), rng0.uniform(-0.5, 0.5), rng0.uniform(-np.pi, np.pi), rng0.uniform(-2, 2)], np.float32)

    step_fn = sharded.step if sharded is not None else eng.step
    if pool is None:
        ptrs = [None] * 16

    for i in range(args.warmup):
        plant_step(s, step_fn(s, ptrs[i & 15])[0])
    # dispatch-timestamp timing of the dominant kernel on a sparse sample of the timed launches: timing a
    # launch costs ~8 us of host time (measured), so timing all of them would distort the metric
    prof_every = 1 if args.steps < 40 else 16
    eng.profile_enable(True, every=prof_every)
    if world > 1 or force_pg:
        dist.barrier()
    torch.cuda.synchronize()
    per_step = np.empty(args.steps)
    t0 = time.perf_counter()
    ta = t0
    for i in range(args.steps):
        plant_step(s, step_fn(s, ptrs[i & 15])[0])        # controller.step, then the plant: closed loop
        tb = time.perf_counter(); per_step[i] = tb - ta; ta = tb
    torch.cuda.synchronize()
    if world > 1 or force_pg:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = eng.profile_read()
    eng.profile_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1 or force_pg:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_units = N * H * world * args.steps
        kms = float(np.mean(kern_ms)) if len(kern_ms) else float("nan")
        alg_bytes, alg_flops = algorithmic(w, P, samples_in_hbm)
        ok = kms == kms and kms > 0
        if alg_flops is not None:
            ach = alg_flops / (kms * 1e-3) / 1e12 if ok else None
            roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / MFMA_F32_PEAK_TF if ach else None, "traffic": None, "algorithmic_flops": alg_flops,
                    "note": "fp32-input MFMA (v_mfma_f32_16x16x4_f32) for exact-fp32 parity; "
                            + ("16 trajectories per workgroup, the GRU step split over its 4 waves (ctk_gru.h); latency-bound: "
                               "64 workgroups on 256 CUs at this size" if w["pred"] == "GRU" else "16 trajectories per wave")}
        else:
            ach = alg_bytes / (kms * 1e-3) / 1e9 if ok else None
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS if ach else None,
                    "traffic": 514891 if args.workload == "mppi_cfg2" and samples_in_hbm else None,
                    "note": "issue/latency-bound at this size, not HBM-bound: 0.2 MB per launch vs an H-step dependent "
                            "recurrence (~60 VALU instructions per step on one wave per 64 trajectories); traffic = "
                            "(2 x FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes with the guide's gfx950 factor on "
                            "FETCH_SIZE (profiles/): 200 KiB of samples + ~270 KiB of per-launch fixed fetches (each of the 8 XCDs' "
                            "L2 starts cold: kernel code, arguments, tables) + record polling, DESIGN.md 5"}
        roof.update({"kernel": eng.dominant_kernel(), "kernel_us": kms * 1e3, "algorithmic_bytes": alg_bytes,
                     "kernel_timed_launches": int(len(kern_ms)), "kernel_timed_every": prof_every})
        ps = per_step * 1e3
        out = {
            "metric": "trajectory-steps/sec (N*H per controller.step)", "value": total_units / elapsed,
            "unit": "trajectory-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w['opt'].upper()} N={N} per GPU, H={H}, period={p}, predictor {w['pred']} "
                                   f"(4 states, 1 input) [{args.workload}]",
                       "samples": args.samples if w["opt"] == "mppi" else "device-rng", "global_rollouts": N * world,
                       "parallelism": ((f"rollout shards x{world}, records of {P + 2} floats exchanged by peer-to-peer stores over xGMI"
                                        if sharded.exchange == "p2p" else
                                        f"rollout shards x{world}, 1 all-gather of {P + 2} floats per step (RCCL)") if world > 1
                                       else "single GPU")},
            "step_ms_median": float(np.median(ps)), "step_ms_p95": float(np.percentile(ps, 95)),
            "roofline": roof,
        }
        if not args.no_large_n and world == 1 and args.workload == "mppi_cfg2":
            out["roofline_large_n"] = large_n_point(torch, CtkEngine, dev, H, p)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
            if w["opt"] == "mppi" and w["pred"] == "ODE":
                out["cpu_baseline_torch"] = cpu_baseline_torch(w)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    eng.close()
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
