#!/usr/bin/env python3
"""Where do the fixed costs of bench.py's timed bracket come from?  (a) an idle torch.cuda.synchronize() — before an engine exists, with an
engine (its own HIP stream = one more hardware queue), after steps; (b) the first step after a synchronize against the steady step,
for the engine on its own stream and on torch's current stream.  Diagnostic only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine


def idle_sync(n=200):
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return np.median(t) * 1e6, np.mean(t) * 1e6


x = torch.zeros(1024, device="cuda"); x += 1; torch.cuda.synchronize()
print("idle torch.cuda.synchronize(), torch only        : median %.1f us, mean %.1f us" % idle_sync())
N, H = 1024, 50
s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
for own in (True, False):
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1, seed=1)
    if not own:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
    tag = "own stream" if own else "torch's current stream"
    print(f"[{tag}] idle synchronize with the engine created  : median %.1f us, mean %.1f us" % idle_sync())
    noise = torch.randn((N, H, 1), device="cuda"); torch.cuda.synchronize()
    for _ in range(50):
        e.step(s, noise.data_ptr())
    print(f"[{tag}] idle synchronize after 50 steps           : median %.1f us, mean %.1f us" % idle_sync())
    first, steady, close = [], [], []
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e.step(s, noise.data_ptr()); t1 = time.perf_counter()
        for _ in range(19):
            e.step(s, noise.data_ptr())
        t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        first.append(t1 - t0); steady.append((t2 - t1) / 19); close.append(t3 - t2)
    print(f"[{tag}] first step after synchronize %.1f us | steady step %.1f us | closing synchronize %.1f us  (medians of 30 brackets of 20 steps)"
          % (np.median(first) * 1e6, np.median(steady) * 1e6, np.median(close) * 1e6))
    # a stream-level wait instead of the device-wide one, for comparison only (the bench contract says torch.cuda.synchronize)
    st = []
    for rep in range(30):
        for _ in range(5):
            e.step(s, noise.data_ptr())
        t0 = time.perf_counter(); torch.cuda.current_stream().synchronize(); st.append(time.perf_counter() - t0)
    print(f"[{tag}] torch current_stream().synchronize() idle   : median %.1f us" % (np.median(st) * 1e6))
    e.close()
