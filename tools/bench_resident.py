#!/usr/bin/env python3
"""Step time of the resident form against the launched form at the headline size (MPPI N 1024, H 50, analytic predictor), CtkEngine.step
with device sample buffers and with the in-kernel sampler; median / mean of 2000 steps each.  (bench.py reports the same in its
`resident` field at the controller_mpc boundary.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine

N, H = 1024, 50
pool = [torch.randn((N * H,), device="cuda") for _ in range(16)]
ptrs = [t.data_ptr() for t in pool]
s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
for resident in (False, True):
    for buf in (True, False):
        e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1, seed=1)
        if resident:
            e.resident_enable(True, idle_us=500.0)
        t = np.empty(2000)
        for i in range(2100):
            s[0] = 0.05 + 0.01 * (i % 7)
            t0 = time.perf_counter()
            e.step(s, ptrs[i & 15] if buf else None)
            if i >= 100:
                t[i - 100] = time.perf_counter() - t0
        st = e.resident_stats()
        print(f"{'resident' if resident else 'launched'} / {'buffer' if buf else 'device-rng'}: median {np.median(t) * 1e6:6.2f} us  mean {t.mean() * 1e6:6.2f}  p99 {np.percentile(t, 99) * 1e6:6.2f}"
              f"   (resident launches {st['launches']}, steps {st['steps']})")
        e.close()
