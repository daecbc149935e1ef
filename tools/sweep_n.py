#!/usr/bin/env python3
"""Scaled-N sweep of the MPPI rollout kernel (where does the kernel stop being
latency-bound?).  N = 2^10 .. 2^22, H = 50, period 1, ODE; sample buffer resident in HBM.
Prints one row per N: kernel time (dispatch timestamps), trajectory-steps/s, algorithmic GB/s and
its fraction of the 8 TB/s HBM peak, the kernel that ran, and the VALU instruction rate implied by ~65 (4-wave kernel) / ~75
(throughput kernels) instructions per trajectory-step against the SIMD-32 issue peak (a wave64 VALU op issues in 2 cycles when
several waves share the SIMD; measured 2.67 on independent FMAs, tools/diag_pk_rate.hip)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine

H = 50
s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
print(f"{'N':>9s} {'blocks':>7s} {'kernel_us':>10s} {'step_us':>9s} {'traj-steps/s':>13s} {'alg GB/s':>9s} {'HBM frac':>9s} {'VALU lane-inst/s':>17s} {'VALU frac':>9s}  kernel")
VALU_PEAK = 256 * 4 * 32 * 2.4e9   # fp32 VALU lanes per second, chip-wide: 32 lanes per SIMD and cycle (157 TFLOP/s of FMA)
for lg in range(10, 23, 2):
    N = 1 << lg
    eng = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1, seed=1)
    noise = torch.randn((N, H, 1), device="cuda")
    ptr = noise.data_ptr()
    for _ in range(5):
        eng.step(s, ptr)
    eng.profile_enable(True, every=1)
    n = 30 if lg <= 18 else 10
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step(s, ptr)
    wall = (time.perf_counter() - t0) / n
    k = float(np.mean(eng.profile_read())) * 1e-3
    alg = 4 * N * H + 4 * N + 8 * H + 16
    rate = N * H / wall
    valu = N * H * (75 if N >= 32768 else 65) / k
    print(f"{N:9d} {(N + 63) // 64:7d} {k * 1e6:10.1f} {wall * 1e6:9.1f} {rate:13.3e} {alg / k / 1e9:9.1f} {alg / k / 8e12:9.4f} {valu:17.3e} {valu / VALU_PEAK:9.3f}  {eng.dominant_kernel()}")
    eng.close(); del noise
    torch.cuda.empty_cache()
