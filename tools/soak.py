#!/usr/bin/env python3
"""Closed-loop soak (GPU box): 3000 steps of MPPI / MPPI+logging / CEM / RPGD / random-action (+ MPPI and RPGD with the MLP predictor) against a host plant with
API calls interleaved (parameters, reset, state round trip, log reads); every output must stay finite and inside the limits.
usage: python tools/soak.py"""
import numpy as np, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from control_toolkit_amd import CtkEngine
from bench import plant_step
rng = np.random.default_rng(0)
engs = {
 "mppi": CtkEngine("mppi", "ODE", num_rollouts=1024, mpc_horizon=50, dt=0.02, seed=1),
 "mppi_log": CtkEngine("mppi", "ODE", num_rollouts=300, mpc_horizon=20, dt=0.02, seed=2, materialize_trajectories=True, period_interpolation_inducing_points=4),
 "cem": CtkEngine("cem", "ODE", num_rollouts=500, mpc_horizon=25, dt=0.02, seed=3, cem_outer_it=3, cem_best_k=50, cem_initial_action_stdev=0.5, cem_stdev_min=0.01),
 "rpgd": CtkEngine("rpgd", "ODE", num_rollouts=64, mpc_horizon=30, dt=0.02, seed=4, outer_its=3, resamp_per=5, shift_previous=1, opt_keep_k=16, sampling_distribution=0, sample_min=-1.0, sample_max=1.0, learning_rate=0.05, gradmax_clip=5.0, period_interpolation_inducing_points=5),
 "rand": CtkEngine("random_action", "ODE", num_rollouts=320, mpc_horizon=35, dt=0.02, seed=5),
 # the MLP predictor: pair form of the MPPI kernel (N <= 8192), wide form of the RPGD descent (phase + Jacobian launches)
 "mppi_mlp": CtkEngine("mppi", "MLP", num_rollouts=1000, mpc_horizon=30, dt=0.02, seed=6, period_interpolation_inducing_points=5),
 "rpgd_mlp": CtkEngine("rpgd", "MLP", num_rollouts=72, mpc_horizon=25, dt=0.02, seed=7, outer_its=3, resamp_per=5, shift_previous=1, opt_keep_k=18, sampling_distribution=0, sample_min=-1.0, sample_max=1.0, learning_rate=0.05, gradmax_clip=5.0, period_interpolation_inducing_points=5),
}
_w = (np.random.default_rng(11).standard_normal(engs["mppi_mlp"].predictor_weight_count()) * 0.15).astype(np.float32)
engs["mppi_mlp"].set_predictor_weights(_w); engs["rpgd_mlp"].set_predictor_weights(_w)
engs["rpgd"].reset(); engs["rpgd_mlp"].reset()
engs["mppi_log"].log_enable(64)
states = {k: np.array([0.0, 0.0, 3.0, 0.0], np.float32) for k in engs}
t0 = time.time()
for i in range(3000):
    for k, e in engs.items():
        u = e.step(states[k])
        assert np.isfinite(u).all() and abs(float(u[0])) <= 1.0 + 1e-6, (k, i, u)
        plant_step(states[k], float(u[0]))
        if not np.isfinite(states[k]).all() or abs(states[k][0]) > 50:
            states[k] = np.array([0.0, 0.0, rng.uniform(-3, 3), 0.0], np.float32)
    if i % 97 == 0:
        engs["mppi"].set_param("target_position", float(rng.uniform(-0.2, 0.2)))
        st = engs["cem"].get_state(); engs["cem"].set_state(st)
    if i % 501 == 0:
        engs["mppi"].reset(); engs["rand"].reset()
        a = engs["mppi_log"].log_read("J", max(0, engs["mppi_log"].log_count() - 10), min(10, engs["mppi_log"].log_count()))
        assert np.isfinite(a).all()
print("soak ok", time.time() - t0, "s;", {k: [round(float(x), 3) for x in v] for k, v in states.items()})
for e in engs.values(): e.close()
