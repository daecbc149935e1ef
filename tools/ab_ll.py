#!/usr/bin/env python3
"""A/B of the in-launch record hand-off: {value, seq} polling (default) vs ticket + fetch (CTK_NO_LL=1), same process,
alternating rounds.  usage: python tools/ab_ll.py [N] [H]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from control_toolkit_amd import CtkEngine
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 50
engines = {}
for name, env in (("ll", None), ("ticket", "1")):
    if env: os.environ["CTK_NO_LL"] = env
    else: os.environ.pop("CTK_NO_LL", None)
    engines[name] = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, seed=1)
pool = [torch.randn((N, H, 1), device="cuda") for _ in range(16)]
ptrs = [p.data_ptr() for p in pool]
torch.cuda.synchronize()
s = np.array([0.0, 0.0, 0.3, 0.0], np.float32)
res = {k: [] for k in engines}
for rnd in range(6):
    for name, e in engines.items():
        for i in range(200):
            e.step(s, ptrs[i & 15])
        t0 = time.perf_counter()
        for i in range(3000):
            e.step(s, ptrs[i & 15])
        res[name].append((time.perf_counter() - t0) / 3000 * 1e6)
for name, v in res.items():
    e = engines[name]
    e.profile_enable(True, every=4)
    for i in range(400):
        e.step(s, ptrs[i & 15])
    k = np.mean(e.profile_read()) * 1e3
    print(f"{name:7s} step us per round: {' '.join(f'{x:6.2f}' for x in v)} | median {np.median(v):6.2f} | kernel {k:6.2f} us")
