#!/usr/bin/env python3
"""bench.py's bracket with the CONTROLLER boundary: per-step times of the first steps after the opening synchronize (30 brackets of 20 steps),
with and without a host-side pause in place of the synchronize.  Diagnostic only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = bench.WORKLOADS["mppi_cfg2"] if hasattr(bench, "WORKLOADS") else None
ctrl = bench.build_controller(w, w["N"], 0, 0, "device")
eng = ctrl.optimizer.engine
N, H = w["N"], w["H"]
pool = [torch.randn((N * H,), device="cuda") for _ in range(16)]
ptrs = [t.data_ptr() for t in pool]
from control_toolkit_amd.others.globals_and_utils import DeviceBufferRng
ctrl.optimizer.rng = DeviceBufferRng(ptrs, seed=1)
s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
for i in range(100):
    bench.plant_step(s, np.asarray(ctrl.step(s)).reshape(-1)[0])
for label, opener in (("torch.cuda.synchronize()", torch.cuda.synchronize), ("time.sleep(100 us), no synchronize", lambda: time.sleep(1e-4)),
                      ("nothing", lambda: None)):
    per = np.zeros((30, 20))
    for rep in range(30):
        opener()
        ta = time.perf_counter()
        for i in range(20):
            bench.plant_step(s, np.asarray(ctrl.step(s)).reshape(-1)[0])
            tb = time.perf_counter(); per[rep, i] = tb - ta; ta = tb
        torch.cuda.synchronize()
    med = np.median(per, axis=0) * 1e6
    print("   bracket 0:", np.round(per[0, :8] * 1e6, 1), " bracket 1:", np.round(per[1, :8] * 1e6, 1), " bracket 15:", np.round(per[15, :8] * 1e6, 1))
    print(f"opener = {label}: step 0 {med[0]:.1f} us, step 1 {med[1]:.1f}, step 2 {med[2]:.1f}, steps 3.. {np.median(med[3:]):.1f}")

# kernel time (dispatch timestamps) against wall time of the steps around a synchronize: is the first step's excess inside the kernel?
eng.profile_enable(True, every=1)
walls, kerns = [], []
for rep in range(20):
    torch.cuda.synchronize()
    w = []
    for i in range(6):
        t0 = time.perf_counter(); ctrl.step(s); w.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    k = np.asarray(eng.profile_read(), np.float64).ravel()
    walls.append(w); kerns.append(k[-6:] if k.size >= 6 else np.full(6, np.nan))
print("with per-launch timing on: wall us (median over 20 brackets) of steps 0..5:", np.round(np.median(np.array(walls), 0) * 1e6, 1))
print("                           kernel us of the same launches               :", np.round(np.nanmedian(np.array(kerns), 0) * 1e-3, 1))
