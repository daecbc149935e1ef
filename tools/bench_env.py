#!/usr/bin/env python3
"""Step time of the environment-agnostic template kernels (csrc/ctk_generic.hip, ctk_generic_net.hip) next to the
hand-tuned CartPole kernels, at the headline problem sizes.  Not a bench.py line (BASELINE.json's metric is quoted on
CartPole); the table goes to profiles/ so DESIGN.md can say what the second environment and the generic path cost.

    python tools/bench_env.py [--steps 300]

Every row: the optimizer step through CtkEngine.step with the in-kernel sampler (no sample tensors), state changing every
call, median and mean of `--steps` calls after 30 untimed ones."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS = [
    # label, env, generic_kernels, optimizer, predictor, N, H, p, kwargs
    ("mppi  ODE", "mppi", "ODE", 1024, 50, 1, {}),
    ("mppi  ODE interp", "mppi", "ODE", 1024, 50, 10, {}),
    ("mppi  MLP", "mppi", "MLP", 1024, 50, 1, {}),
    ("mppi  MLP shard", "mppi", "MLP", 8192, 100, 10, {}),
    ("mppi  GRU", "mppi", "GRU", 1024, 50, 1, {}),
    ("cem   ODE", "cem", "ODE", 4096, 30, 1, dict(cem_outer_it=3, cem_best_k=409)),
    ("rand  ODE", "random_action", "ODE", 4096, 30, 1, {}),
    ("rpgd  ODE", "rpgd", "ODE", 256, 50, 10, dict(outer_its=10, resamp_per=10, opt_keep_k=64, sampling_distribution=0)),
    ("rpgd  MLP", "rpgd", "MLP", 256, 50, 10, dict(outer_its=10, resamp_per=10, opt_keep_k=64, sampling_distribution=0)),
    ("rpgd  GRU", "rpgd", "GRU", 256, 50, 10, dict(outer_its=10, resamp_per=10, opt_keep_k=64, sampling_distribution=0)),
    # hidden widths 33..64: the 64-unit form of the one-wave template kernels (csrc/ctk_mlp_wide.h) for every environment, CartPole included
    ("mppi  MLP h64", "mppi", "MLP", 1024, 50, 1, dict(predictor_hidden=(64, 64))),
    ("mppi  MLP h64 shard", "mppi", "MLP", 8192, 100, 10, dict(predictor_hidden=(64, 64))),
    ("rpgd  MLP h64", "rpgd", "MLP", 256, 50, 10, dict(outer_its=10, resamp_per=10, opt_keep_k=64, sampling_distribution=0, predictor_hidden=(64, 64))),
]


def weights(eng, seed=0):
    n = eng.predictor_weight_count()
    if n:
        rng = np.random.default_rng(seed)
        eng.set_predictor_weights((rng.standard_normal(n) * 0.15).astype(np.float32))


def run(env, generic, opt, pred, N, H, p, kw, steps):
    import torch
    from control_toolkit_amd import CtkEngine
    e = CtkEngine(opt, pred, environment=env, generic_kernels=generic, num_rollouts=N, mpc_horizon=H, dt=0.02,
                  period_interpolation_inducing_points=p, seed=1, **kw)
    weights(e)
    if opt == "rpgd":
        e.reset(None)
    s = np.zeros(e.S, np.float32)
    s[:min(4, e.S)] = [0.1, 0.0, 0.2, 0.0][:min(4, e.S)]
    t = np.empty(steps)
    for i in range(30 + steps):
        s[0] = 0.1 + 0.01 * (i % 7)
        t0 = time.perf_counter()
        e.step(s, None)
        if i >= 30:
            t[i - 30] = time.perf_counter() - t0
    torch.cuda.synchronize()
    name = e.kernel_name() if hasattr(e, "kernel_name") else ""
    e.close()
    return float(np.median(t) * 1e6), float(t.mean() * 1e6), name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rows", default="", help="only the rows whose label contains this text")
    a = ap.parse_args()
    print(f"{'workload':<18}{'N':>6}{'H':>5}{'p':>4} | {'CartPole tuned':>16} | {'CartPole template':>18} | {'Quad2D template':>16} | {'Hover template':>16}   (us / step, median (mean))")
    for label, opt, pred, N, H, p, kw in ROWS:
        if a.rows and a.rows not in label:
            continue
        cells = []
        for env, generic in (("CartPole", False), ("CartPole", True), ("Quad2D", False), ("Hover", False)):
            try:
                med, mean, _ = run(env, generic, opt, pred, N, H, p, kw, a.steps)
                cells.append(f"{med:8.1f} ({mean:6.1f})")
            except Exception as ex:                                  # a combination the library refuses is part of the table
                cells.append(f"refused: {type(ex).__name__}")
        print(f"{label:<18}{N:>6}{H:>5}{p:>4} | {cells[0]:>16} | {cells[1]:>18} | {cells[2]:>16} | {cells[3]:>16}", flush=True)


if __name__ == "__main__":
    main()
