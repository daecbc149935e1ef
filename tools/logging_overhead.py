#!/usr/bin/env python3
"""Cost of optimizer_logging (reference optimizer_mppi.py:214-218: Q, J, rollout trajectories to the host every
step) at the BASELINE MPPI size.  usage: python tools/logging_overhead.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from control_toolkit_amd import CtkEngine

N, H = 1024, 50
s = np.array([0.0, 0.0, 0.3, 0.0], np.float32)
for log in (False, True):
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, materialize_trajectories=log, seed=1)
    for _ in range(50):
        e.step(s)
    t0 = time.perf_counter()
    for _ in range(500):
        e.step(s)
    t_step = (time.perf_counter() - t0) / 500
    line = f"materialize={log}: step {t_step*1e6:7.1f} us"
    if log:
        for name in ("Q", "J", "TRAJ"):
            t0 = time.perf_counter()
            for _ in range(200):
                a = e.read(name)
            dt = (time.perf_counter() - t0) / 200
            line += f" | read {name} {a.nbytes/1024:.0f} KiB {dt*1e6:6.1f} us"
        t0 = time.perf_counter()
        for _ in range(200):
            e.step(s); e.read("Q"); e.read("J"); e.read("TRAJ")
        line += f" | step + 3 reads {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us"
        e.log_enable(256)
        for _ in range(20):
            e.step(s)
        t0 = time.perf_counter()
        for _ in range(200):
            e.step(s)
        line += f" | step with the HBM log ring {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us"
        t0 = time.perf_counter()
        a = e.log_read("TRAJ", e.log_count() - 200, 200); b = e.log_read("Q", e.log_count() - 200, 200); c = e.log_read("J", e.log_count() - 200, 200)
        line += f" | bulk read of 200 steps {(time.perf_counter() - t0) * 1e3:6.1f} ms ({(a.nbytes + b.nbytes + c.nbytes) / 2**20:.0f} MiB)"
    print(line)
    e.close()
