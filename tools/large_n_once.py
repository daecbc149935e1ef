#!/usr/bin/env python3
"""A few MPPI steps at N = 2^20 (H = 50, period 1, ODE, sample buffer in HBM) and nothing else: the target of the SQ_INSTS_VALU pass of
tools/gpu_run_profiles.sh (instructions per trajectory-step of ctk_mppi_rollout_tps = SQ_INSTS_VALU / (N / 64 * H))."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine

N, H = 1 << 20, 50
eng = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1, seed=1)
noise = torch.randn((N, H, 1), device="cuda")
s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
for _ in range(6):
    eng.step(s, noise.data_ptr())
torch.cuda.synchronize()
print(eng.dominant_kernel())
eng.close()
