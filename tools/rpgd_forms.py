#!/usr/bin/env python3
"""RPGD + MLP on CartPole's own kernels, per MPC step, over population sizes and horizons: the one-launch form (ctk_rpgd_mlp_persistent) or —
with CTK_RPGD_NO_PERSISTENT=1 in the environment (the switch is read once per process) — the phase launches it replaces.
usage: python tools/rpgd_forms.py; CTK_RPGD_NO_PERSISTENT=1 python tools/rpgd_forms.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from control_toolkit_amd import CtkEngine
for N, H, its in ((64, 30, 10), (128, 50, 20), (256, 50, 20), (512, 50, 20), (512, 64, 5), (96, 20, 3)):
    K = N // 4
    e = CtkEngine("rpgd", "MLP", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=10, outer_its=its, resamp_per=10,
                  opt_keep_k=K, seed=3)
    e.set_predictor_weights((np.random.default_rng(0).standard_normal(e.predictor_weight_count()) * 0.15).astype(np.float32))
    e.reset()
    s = np.array([0.05, 0.0, 2.9, 0.3], np.float32)
    for _ in range(10): e.step(s)
    t0 = time.perf_counter()
    for _ in range(60): e.step(s)
    print(f"N {N} H {H} its {its}: {e.dominant_kernel()}: {(time.perf_counter() - t0) / 60 * 1e6:.1f} us per step", flush=True)
    e.close()
