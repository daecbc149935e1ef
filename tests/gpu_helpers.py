"""Helpers for the -m gpu parity tests: build engines (through the C ABI) from a golden fixture."""
import numpy as np

from control_toolkit_amd import CtkEngine
from helpers import env_from, rpgd_kwargs_from

ENV_NAMES = ("g", "m_cart", "m_pole", "L", "u_max", "M_fric", "J_fric", "target_position", "target_equilibrium",
             "dd_weight", "ep_weight", "ekp_weight", "cc_weight", "ccrc_weight", "R", "x_scale", "terminal_weight")


def apply_env(engine: CtkEngine, env):
    for n in ENV_NAMES:
        engine.set_param(n, float(getattr(env, n)))


def mppi_engine_from(d, materialize=True, **kw) -> CtkEngine:
    e = CtkEngine("mppi", str(d["predictor"]), num_rollouts=int(d["num_rollouts"]), mpc_horizon=int(d["mpc_horizon"]),
                  dt=float(d["dt"]), action_low=float(d["low"][0]), action_high=float(d["high"][0]),
                  period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]),
                  materialize_trajectories=materialize, cc_weight=float(d["cc_weight"]), R=float(d["R"]),
                  LBD=float(d["LBD"]), NU=float(d["NU"]), SQRTRHOINV=float(d["SQRTRHOINV"]), **kw)
    apply_env(e, env_from(d))
    if str(d["predictor"]) == "MLP":
        e.set_predictor_weights(d["mlp_weights"])
    return e


def rpgd_engine_from(d, materialize=False, **kw) -> CtkEngine:
    k = rpgd_kwargs_from(d)
    N = int(d["num_rollouts"])
    lo, hi = float(d["low"][0]), float(d["high"][0])
    smin, smax = (lo, hi) if k["sample_whole_control_space"] else (k["uniform_dist_min"], k["uniform_dist_max"])
    e = CtkEngine("rpgd", str(d["predictor"]), num_rollouts=N, mpc_horizon=int(d["mpc_horizon"]), dt=float(d["dt"]),
                  action_low=lo, action_high=hi,
                  period_interpolation_inducing_points=k["period_interpolation_inducing_points"],
                  materialize_trajectories=materialize, outer_its=k["outer_its"], resamp_per=k["resamp_per"],
                  shift_previous=k["shift_previous"], opt_keep_k=int(max(int(N * k["opt_keep_k_ratio"]), 1)),
                  sampling_distribution=0 if k["SAMPLING_DISTRIBUTION"] == "uniform" else 1,
                  sample_stdev=k["sample_stdev"], sample_mean=k["sample_mean"], sample_min=smin, sample_max=smax,
                  learning_rate=k["learning_rate"], gradmax_clip=k["gradmax_clip"], adam_beta_1=k["adam_beta_1"],
                  adam_beta_2=k["adam_beta_2"], adam_epsilon=k["adam_epsilon"],
                  warmup=int(k["warmup"]), warmup_iterations=k["warmup_iterations"], **kw)
    apply_env(e, env_from(d))
    if str(d["predictor"]) == "MLP":
        e.set_predictor_weights(d["mlp_weights"])
    return e
