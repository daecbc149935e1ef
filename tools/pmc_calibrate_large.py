#!/usr/bin/env python3
"""FETCH_SIZE of the MPPI step as a function of N (known sample bytes 200 * N): separates the per-launch fixed fetches
(code, arguments, tables, record hand-off) from the sample stream, and shows the counter's factor on that stream.
usage: rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o cal -- python3 tools/pmc_calibrate_large.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from control_toolkit_amd import CtkEngine
dev = torch.device("cuda", 0)
s = np.array([0.0, 0.0, 0.3, 0.0], np.float32)
for N in (64, 128, 256, 512, 1024, 2048, 4096, 16384, 1 << 20):
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=50, dt=0.02, seed=1)
    pool = [torch.randn((N, 50, 1), device=dev) for _ in range(4 if N > 100000 else 16)]
    for i in range(32):
        e.step(s, pool[i % len(pool)].data_ptr())
    e.close()
    del pool
