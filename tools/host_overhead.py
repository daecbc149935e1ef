#!/usr/bin/env python3
"""Where does a step() spend its wall time?  (diagnostic; not a benchmark)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine

def timeit(fn, n=300, warm=30):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e6
    return f"median {np.median(ts):7.2f} us  p10 {np.percentile(ts,10):7.2f}  p90 {np.percentile(ts,90):7.2f}"

s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
for (N, H) in [(64, 1), (1024, 50)]:
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1, seed=1)
    noise = torch.randn((N, H, 1), device="cuda")
    ptr = noise.data_ptr()
    print(f"MPPI N={N} H={H}: device buffer  ", timeit(lambda: e.step(s, ptr)))
    print(f"MPPI N={N} H={H}: device rng     ", timeit(lambda: e.step(s, None)))
    e.profile_enable(True)
    for _ in range(50): e.step(s, ptr)
    k = e.profile_read()
    print(f"   rollout kernel (dispatch timestamps): mean {k.mean()*1e3:.2f} us")
    e.close()
import ctypes
print("python no-op ctypes call:", timeit(lambda: e._lib.ctk_abi_version()))
