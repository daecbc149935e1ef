"""Shared helpers for the parity tests (oracle construction from a golden fixture)."""
import os

import numpy as np

from oracle import ctk_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def env_from(d):
    """the environment a fixture was recorded on (CartPole unless the file says otherwise)"""
    names = [str(x) for x in d["env_param_names"]]
    cls = O.ENVIRONMENTS[str(d["environment"])] if "environment" in d.files else O.EnvParams
    return cls(**{n: float(v) for n, v in zip(names, d["env_params"])})


def predictor_from(d) -> O.Predictor:
    hidden = tuple(int(x) for x in d["hidden_sizes"]) if "hidden_sizes" in d.files else (32, 32)
    return O.Predictor(kind=str(d["predictor"]), dt=float(d["dt"]), env=env_from(d), weights=d["mlp_weights"] if "mlp_weights" in d.files else None,
                       hidden_sizes=hidden)


def mppi_oracle_from(d) -> O.MPPI:
    pred = predictor_from(d)
    return O.MPPI(pred, O.Cost(pred.env, pred.dt), d["low"], d["high"],
                  num_rollouts=int(d["num_rollouts"]), mpc_horizon=int(d["mpc_horizon"]),
                  cc_weight=float(d["cc_weight"]), R=float(d["R"]), LBD=float(d["LBD"]), NU=float(d["NU"]),
                  SQRTRHOINV=float(d["SQRTRHOINV"]),
                  period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]))


RPGD_KEYS = ("outer_its", "sample_stdev", "sample_mean", "sample_whole_control_space", "uniform_dist_min",
             "uniform_dist_max", "resamp_per", "period_interpolation_inducing_points", "SAMPLING_DISTRIBUTION",
             "shift_previous", "warmup", "warmup_iterations", "learning_rate", "opt_keep_k_ratio", "gradmax_clip",
             "adam_beta_1", "adam_beta_2", "adam_epsilon")


def rpgd_kwargs_from(d) -> dict:
    kw = {}
    for k in RPGD_KEYS:
        v = d[k]
        if v.dtype.kind in "US":
            kw[k] = str(v)
        elif v.dtype.kind == "b":
            kw[k] = bool(v)
        elif v.dtype.kind in "iu":
            kw[k] = int(v)
        else:
            kw[k] = float(v)
    return kw


def rpgd_oracle_from(d) -> O.RPGD:
    pred = predictor_from(d)
    return O.RPGD(pred, O.Cost(pred.env, pred.dt), d["low"], d["high"],
                  num_rollouts=int(d["num_rollouts"]), mpc_horizon=int(d["mpc_horizon"]), **rpgd_kwargs_from(d))


def cem_oracle_from(d) -> O.CEM:
    pred = predictor_from(d)
    return O.CEM(pred, O.Cost(pred.env, pred.dt), d["low"], d["high"], num_rollouts=int(d["num_rollouts"]),
                 mpc_horizon=int(d["mpc_horizon"]), cem_outer_it=int(d["cem_outer_it"]),
                 cem_initial_action_stdev=float(d["cem_initial_action_stdev"]), cem_stdev_min=float(d["cem_stdev_min"]),
                 cem_best_k=int(d["cem_best_k"]), warmup=bool(d["warmup"]), warmup_iterations=int(d["warmup_iterations"]))


def random_oracle_from(d) -> O.RandomAction:
    pred = predictor_from(d)
    return O.RandomAction(pred, O.Cost(pred.env, pred.dt), d["low"], d["high"], num_rollouts=int(d["num_rollouts"]),
                          mpc_horizon=int(d["mpc_horizon"]))


def cem_naive_grad_oracle_from(d) -> O.CEMNaiveGrad:
    pred = predictor_from(d)
    return O.CEMNaiveGrad(pred, O.Cost(pred.env, pred.dt), d["low"], d["high"], num_rollouts=int(d["num_rollouts"]),
                          mpc_horizon=int(d["mpc_horizon"]), cem_outer_it=int(d["cem_outer_it"]),
                          cem_initial_action_stdev=float(d["cem_initial_action_stdev"]), cem_stdev_min=float(d["cem_stdev_min"]),
                          cem_best_k=int(d["cem_best_k"]), learning_rate=float(d["learning_rate"]), gradmax_clip=float(d["gradmax_clip"]))


def cut_is_separated(J, K, rel=1e-4):
    """True when the K-th and (K+1)-th smallest costs are further apart than fp32 cost noise: only then is the elite SET (and
    everything refitted from it) comparable between two fp32 evaluations of the same costs"""
    srt = np.sort(np.asarray(J))
    return K >= len(srt) or (srt[K] - srt[K - 1]) > rel * abs(srt[K - 1])


MPPI_CASES = ["tiny_ode", "interp_ode", "cfg2_ode", "quirk_ode", "mlp", "mlp_h16", "mlp_h24_8", "mlp_h64"]   # (the last three: 5-16-16-4, 5-24-8-4 and 5-64-64-4 networks)
RPGD_CASES = ["ode_small", "ode_its20", "mlp_cfg4", "ode_normal"]
# recorded from the same unmodified reference optimizers on the second environment (6 states, 2 control inputs)
MPPI_QUAD_CASES = ["quad2d", "quad2d_p1"]
RPGD_QUAD_CASES = ["quad2d", "quad2d_its20"]
# ... and on the third environment (7 states, 3 control inputs: 10 network inputs), analytic and 10-32-32-7 MLP predictor
MPPI_HOVER_CASES = ["hover_ode", "hover_mlp"]
RPGD_HOVER_CASES = ["hover_ode", "hover_mlp"]
# recorded from the unmodified TF-only optimizers (optimizer_cem_tf, optimizer_random_action_tf, optimizer_cem_naive_grad_tf) through the
# reference's controller_mpc with the torch-backed `tensorflow` stand-in (tests/golden/standins/tensorflow)
CEM_CASES = ["tiny", "default", "cfg3", "warmup", "mlp", "quad2d", "hover", "hover_mlp"]
RANDOM_CASES = ["cfg1", "default", "quad2d", "hover_mlp"]
CEM_NAIVE_GRAD_CASES = ["default", "its2", "hover"]
