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
 # round 3: the one-launch CEM at cfg3's size, the split network kernels of the template path (GRU over four waves forward and BPTT, MLP over two),
 # the second and third environment
engs.update({
 "cem_cfg3": CtkEngine("cem", "ODE", num_rollouts=4096, mpc_horizon=30, dt=0.02, seed=8, cem_outer_it=3, cem_best_k=409),
 "mppi_gru_t": CtkEngine("mppi", "GRU", generic_kernels=True, num_rollouts=256, mpc_horizon=25, dt=0.02, seed=9, period_interpolation_inducing_points=5),
 "rpgd_gru_t": CtkEngine("rpgd", "GRU", num_rollouts=40, mpc_horizon=20, dt=0.02, seed=10, outer_its=2, resamp_per=5, opt_keep_k=10, sampling_distribution=0, period_interpolation_inducing_points=5),
 "mppi_mlp_t": CtkEngine("mppi", "MLP", generic_kernels=True, num_rollouts=500, mpc_horizon=30, dt=0.02, seed=12, period_interpolation_inducing_points=5),
 "rpgd_mlp_t": CtkEngine("rpgd", "MLP", generic_kernels=True, num_rollouts=72, mpc_horizon=25, dt=0.02, seed=13, outer_its=3, resamp_per=5, opt_keep_k=18, sampling_distribution=0, period_interpolation_inducing_points=5),
})
others = {
 "quad_mppi": CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=512, mpc_horizon=30, dt=0.02, seed=14, period_interpolation_inducing_points=5),
 "quad_cem": CtkEngine("cem", "MLP", environment="Quad2D", num_rollouts=256, mpc_horizon=20, dt=0.02, seed=15, cem_outer_it=2, cem_best_k=25),
 "hover_mppi_mlp": CtkEngine("mppi", "MLP", environment="Hover", num_rollouts=256, mpc_horizon=20, dt=0.02, seed=16, period_interpolation_inducing_points=4),
 "hover_rpgd": CtkEngine("rpgd", "ODE", environment="Hover", num_rollouts=48, mpc_horizon=16, dt=0.02, seed=17, outer_its=2, resamp_per=4, opt_keep_k=12, sampling_distribution=0, period_interpolation_inducing_points=4),
}
# round 4: the in-launch merge of 256 narrow records (one rank's share of configs[4]), the 64-unit MLP on the one-wave template kernels, a 16 / 24
# network embedded into the 32-unit kernels, RPGD with materialised trajectories (the logging rollout), a USER environment (tests/envs/pendulum_env.h)
engs.update({
 "mppi_cfg5_shard": CtkEngine("mppi", "MLP", num_rollouts=8192, mpc_horizon=100, dt=0.02, seed=23, period_interpolation_inducing_points=10),
 "mppi_mlp_h64": CtkEngine("mppi", "MLP", num_rollouts=512, mpc_horizon=30, dt=0.02, seed=24, period_interpolation_inducing_points=5, predictor_hidden=(64, 48)),
 "rpgd_mlp_h64": CtkEngine("rpgd", "MLP", num_rollouts=48, mpc_horizon=16, dt=0.02, seed=25, outer_its=2, resamp_per=5, opt_keep_k=12, sampling_distribution=0, period_interpolation_inducing_points=4, predictor_hidden=(64, 64)),
 "mppi_mlp_h16": CtkEngine("mppi", "MLP", num_rollouts=512, mpc_horizon=30, dt=0.02, seed=26, period_interpolation_inducing_points=5, predictor_hidden=(16, 24)),
 "rpgd_log": CtkEngine("rpgd", "ODE", num_rollouts=96, mpc_horizon=20, dt=0.02, seed=27, outer_its=2, resamp_per=3, opt_keep_k=24, sampling_distribution=0, period_interpolation_inducing_points=5, materialize_trajectories=True),
})
assert engs["mppi_cfg5_shard"].dominant_kernel().startswith("ctk_mppi_rollout<0, 3")
from control_toolkit_amd.build_env import register_environment
register_environment(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "envs", "pendulum_env.h"))
# round 4, later: the template's one-launch RPGD descent (Quad2D, Hover: resident Jacobian workers), the ten-input GRU (Hover)
others.update({
 "quad_rpgd_mlp": CtkEngine("rpgd", "MLP", environment="Quad2D", num_rollouts=64, mpc_horizon=20, dt=0.02, seed=31, outer_its=3, resamp_per=4, opt_keep_k=16, sampling_distribution=0, period_interpolation_inducing_points=5),
 "hover_rpgd_mlp": CtkEngine("rpgd", "MLP", environment="Hover", num_rollouts=48, mpc_horizon=16, dt=0.02, seed=32, outer_its=2, resamp_per=4, opt_keep_k=12, sampling_distribution=0, period_interpolation_inducing_points=4),
 "hover_mppi_gru": CtkEngine("mppi", "GRU", environment="Hover", num_rollouts=128, mpc_horizon=15, dt=0.02, seed=33, period_interpolation_inducing_points=5),
})
assert "rpgd_persist" in others["quad_rpgd_mlp"].dominant_kernel() and "SplitGru" in others["hover_mppi_gru"].dominant_kernel()
others_user = {
 "pend_mppi": CtkEngine("mppi", "ODE", environment="Pendulum", num_rollouts=1024, mpc_horizon=40, dt=0.02, seed=28),
 "pend_cem": CtkEngine("cem", "ODE", environment="Pendulum", num_rollouts=512, mpc_horizon=25, dt=0.02, seed=29, cem_outer_it=3, cem_best_k=50),
 "pend_rpgd": CtkEngine("rpgd", "ODE", environment="Pendulum", num_rollouts=64, mpc_horizon=20, dt=0.02, seed=30, outer_its=3, resamp_per=4, opt_keep_k=16, sampling_distribution=0, period_interpolation_inducing_points=5),
}
for k, e in list(engs.items()) + list(others.items()):
    n = e.predictor_weight_count(getattr(e, "predictor_hidden", None))
    if n and k not in ("mppi_mlp", "rpgd_mlp"):
        e.set_predictor_weights((np.random.default_rng(20).standard_normal(n) * 0.15).astype(np.float32))
# the resident form under two regimes: idle time shorter than the gap between its steps here (it leaves and is launched again every step) and longer
engs["mppi_res_short"] = CtkEngine("mppi", "ODE", num_rollouts=1024, mpc_horizon=50, dt=0.02, seed=21)
engs["mppi_res_long"] = CtkEngine("mppi", "ODE", num_rollouts=512, mpc_horizon=30, dt=0.02, seed=22, period_interpolation_inducing_points=5)
engs["mppi_res_short"].resident_enable(True, 50.0)
engs["mppi_res_long"].resident_enable(True, 50000.0)
for k in ("rpgd_gru_t", "rpgd_mlp_t", "rpgd_mlp_h64", "rpgd_log"):
    engs[k].reset()
others["hover_rpgd"].reset(); others["quad_rpgd_mlp"].reset(); others["hover_rpgd_mlp"].reset()
others_user["pend_rpgd"].reset()
others.update(others_user)
ostate = {k: np.zeros(e.S, np.float32) for k, e in others.items()}
for v in ostate.values():
    v[:min(3, v.size)] = [0.1, 0.0, 0.2][:min(3, v.size)]
_w = (np.random.default_rng(11).standard_normal(engs["mppi_mlp"].predictor_weight_count()) * 0.15).astype(np.float32)
engs["mppi_mlp"].set_predictor_weights(_w); engs["rpgd_mlp"].set_predictor_weights(_w)
engs["rpgd"].reset(); engs["rpgd_mlp"].reset()
engs["mppi_log"].log_enable(64)
states = {k: np.array([0.0, 0.0, 3.0, 0.0], np.float32) for k in engs}
t0 = time.time()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
slow = {}
for i in range(STEPS):
    for k, e in engs.items():
        _t0 = time.perf_counter()
        u = e.step(states[k])
        slow[k] = max(slow.get(k, 0.0), time.perf_counter() - _t0) if i > 5 else 0.0
        assert np.isfinite(u).all() and abs(float(u[0])) <= 1.0 + 1e-6, (k, i, u)
        plant_step(states[k], float(u[0]))
        if not np.isfinite(states[k]).all() or abs(states[k][0]) > 50:
            states[k] = np.array([0.0, 0.0, rng.uniform(-3, 3), 0.0], np.float32)
    for k, e in others.items():            # no host plant for these: the state drifts a little every step
        _t0 = time.perf_counter()
        u = e.step(ostate[k])
        slow[k] = max(slow.get(k, 0.0), time.perf_counter() - _t0) if i > 5 else 0.0
        assert np.isfinite(u).all() and (np.abs(u) <= 1.0 + 1e-6).all(), (k, i, u)
        ostate[k][0] = 0.1 + 0.05 * np.sin(0.01 * i)
        if e.S > 2:
            ostate[k][2] = 0.2 + 0.05 * np.cos(0.013 * i)
    if i % 97 == 0:
        engs["mppi"].set_param("target_position", float(rng.uniform(-0.2, 0.2)))
        st = engs["cem"].get_state(); engs["cem"].set_state(st)
    if i % 501 == 0:
        engs["mppi"].reset(); engs["rand"].reset()
        a = engs["mppi_log"].log_read("J", max(0, engs["mppi_log"].log_count() - 10), min(10, engs["mppi_log"].log_count()))
        assert np.isfinite(a).all()
        assert np.isfinite(engs["rpgd_log"].read("TRAJ")).all() and engs["rpgd_log"].read("AGES_LOGGED").min() >= 0
print("slowest step per engine (ms):", {k: round(v * 1e3, 2) for k, v in slow.items()})
print("resident:", {k: engs[k].resident_stats() for k in ("mppi_res_short", "mppi_res_long")})
print("soak ok", time.time() - t0, "s;", {k: [round(float(x), 3) for x in v] for k, v in states.items()})
for e in list(engs.values()) + list(others.values()): e.close()
