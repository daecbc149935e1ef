#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration for the traffic figure (MI355X_MICROARCH.md: 'calibrate on a known byte count in
your own access pattern'): known-size device-to-device copies next to the MPPI step, in one rocprofv3 --pmc pass.
usage: rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o cal -- python3 tools/pmc_calibrate.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine

dev = torch.device("cuda", 0)
for nbytes in (200 * 1024, 8 * 1024 * 1024, 256 * 1024 * 1024):
    x = torch.randn(nbytes // 4, device=dev)
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)                      # 16-B-per-lane streaming read + write of `nbytes`
    torch.cuda.synchronize()
    del x, y
e = CtkEngine("mppi", "ODE", num_rollouts=1024, mpc_horizon=50, dt=0.02, seed=1)
pool = [torch.randn((1024, 50, 1), device=dev) for _ in range(16)]
s = np.array([0.0, 0.0, 0.3, 0.0], np.float32)
for i in range(40):
    e.step(s, pool[i & 15].data_ptr())
e.close()
