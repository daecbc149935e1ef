#!/usr/bin/env python3
"""The in-launch hand-off of the template's wide RPGD descent under a loaded GPU: a second process keeps every CU busy with
MPPI + MLP steps at N = 65 536 while this one runs closed-loop RPGD + MLP steps (CartPole through the template, Quad2D, Hover).  The same
steps run again on the idle GPU must give the same inputs bit for bit: a poll that ran out would have left NaN records, a stale read a
different plan."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LOAD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from control_toolkit_amd import CtkEngine
e = CtkEngine("mppi", "MLP", num_rollouts=65536, mpc_horizon=100, dt=0.02, period_interpolation_inducing_points=10, seed=3)
e.set_predictor_weights((np.random.default_rng(0).standard_normal(e.predictor_weight_count()) * 0.15).astype(np.float32))
s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
import time
t0 = time.time()
n = 0
while time.time() - t0 < float(sys.argv[1]):
    e.step(s); n += 1
print("load steps", n)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ERRORS = []


def run(steps):
    from control_toolkit_amd import CtkEngine
    out = {}
    for name, env, kw, kernel in (("CartPole", "CartPole", dict(generic_kernels=True), "rpgd_"), ("Quad2D", "Quad2D", {}, "rpgd_"),
                                  ("Hover", "Hover", {}, "rpgd_"),       # (the template's one-launch form; CTK_RPGD_NO_PERSISTENT=1: its phase launches)
                                  # CartPole's own kernels: the whole descent in one launch, resident workers, words / records / flags both ways
                                  ("CartPole-1L", "CartPole", {}, "ctk_rpgd_mlp_persistent")):
        e = CtkEngine("rpgd", "MLP", environment=env, num_rollouts=256, mpc_horizon=50, dt=0.02, period_interpolation_inducing_points=10,
                      outer_its=10, resamp_per=10, opt_keep_k=64, seed=5, **kw)
        assert kernel in e.dominant_kernel(), e.dominant_kernel()
        e.set_predictor_weights((np.random.default_rng(0).standard_normal(e.predictor_weight_count()) * 0.15).astype(np.float32))
        e.reset()
        S = e.S
        s = (0.05 * np.arange(1, S + 1)).astype(np.float32)
        us = []
        for t in range(steps):
            try:
                u = np.asarray(e.step(s)).reshape(-1).copy()
            except Exception as ex:                      # a hand-off that timed out: the step says so, the engine stays usable
                print(f"{name} step {t}: {ex}", flush=True)
                ERRORS.append((name, t))
                u = np.zeros(e.C, np.float32)           # (the loop goes on from a defined input)
            us.append(u)
            s = (0.97 * s + 0.02 * np.resize(u, S) + 0.01 * np.sin(0.1 * t + np.arange(S))).astype(np.float32)
        plans = e.read("PLAN").reshape(-1).copy()        # (the inputs saturate in this loop: the final population is the sharper witness)
        e.close()
        out[name] = np.concatenate([np.stack(us).reshape(-1), plans])
    return out


def run_records(steps):
    """The other in-launch hand-offs — block records {value, seq} merged by the launch that produced them: MPPI (records of 8-byte words;
    one rank's share of configs[4] with its 256 narrow records), the one-launch CEM, the template's MPPI — same comparison."""
    from control_toolkit_amd import CtkEngine
    out = {}
    mk = {
        "mppi_cfg3": lambda: CtkEngine("mppi", "ODE", num_rollouts=4096, mpc_horizon=50, dt=0.02, seed=41),
        "mppi_shard": lambda: CtkEngine("mppi", "MLP", num_rollouts=8192, mpc_horizon=100, dt=0.02, seed=42, period_interpolation_inducing_points=10),
        "cem_cfg3": lambda: CtkEngine("cem", "ODE", num_rollouts=4096, mpc_horizon=30, dt=0.02, seed=43, cem_outer_it=3, cem_best_k=409),
        "quad_mppi": lambda: CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=2048, mpc_horizon=30, dt=0.02, seed=44, period_interpolation_inducing_points=5),
    }
    def resident():
        e = CtkEngine("mppi", "ODE", num_rollouts=1024, mpc_horizon=50, dt=0.02, seed=45)
        e.resident_enable(True, 50000.0)                 # the mailbox kernel stays on the GPU between steps, beside the other process
        return e
    mk["mppi_resident"] = resident
    for name, make in mk.items():
        e = make()
        n = e.predictor_weight_count()
        if n: e.set_predictor_weights((np.random.default_rng(0).standard_normal(n) * 0.15).astype(np.float32))
        s = (0.05 * np.arange(1, e.S + 1)).astype(np.float32)
        us = []
        for t in range(steps):
            try:
                u = np.asarray(e.step(s)).reshape(-1).copy()
            except Exception as ex:
                print(f"{name} step {t}: {ex}", flush=True)
                ERRORS.append((name, t))
                u = np.zeros(e.C, np.float32)
            us.append(u)
            s = (0.97 * s + 0.02 * np.resize(u, e.S) + 0.01 * np.sin(0.1 * t + np.arange(e.S))).astype(np.float32)
        e.close()
        out[name] = np.stack(us)
    return out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    t0 = time.time()
    idle = run(steps)
    t_idle = time.time() - t0
    load = subprocess.Popen([sys.executable, "-c", LOAD, str(8 * t_idle + 40)], stdout=subprocess.PIPE, text=True)
    time.sleep(8)                                # the load process has created its engine and is stepping
    t0 = time.time()
    busy = run(steps)
    t_busy = time.time() - t0
    busy_rec = run_records(steps)
    load.terminate(); load.wait()
    idle_rec = run_records(steps)
    ok = True
    for name in idle_rec:
        same = np.array_equal(idle_rec[name], busy_rec[name]) and np.isfinite(busy_rec[name]).all()
        ok &= same
        print(f"{name:10s} {steps} steps, block records merged in the launch: idle GPU vs loaded GPU {'identical' if same else 'DIFFERENT'}; last u {busy_rec[name][-1]}")
    for env in idle:
        same = np.array_equal(idle[env], busy[env]) and np.isfinite(busy[env]).all()
        ok &= same
        print(f"{env:11s} {steps} steps x 10 iterations handed off in-launch: idle GPU vs loaded GPU {'identical' if same else 'DIFFERENT'} (inputs of every step + the final population, bit for bit)")
    print(f"wall: idle {t_idle:.1f} s, under load {t_busy:.1f} s")
    ok &= not ERRORS
    print(f"steps that reported an error: {len(ERRORS)}")
    print("soak_handoff", "ok" if ok else "FAILED")
    sys.exit(0 if ok else 1)
