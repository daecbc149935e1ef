#!/usr/bin/env python3
"""Build container only (needs /root/reference).  DESIGN.md section 3 makes ONE semantic choice the reference leaves to its
un-vendored SI_Toolkit: `lib.assign(variable, value)`.  The golden fixtures were recorded with value (TensorFlow) semantics —
the variable is rebound, earlier slices of it stay independent tensors.  This probe measures what that choice decides: it drives
the UNMODIFIED reference optimizer_rpgd through controller_mpc twice on the same seeds, once with the value-semantics stand-in
and once with a torch in-place `copy_`, and reports which of the step's outputs differ.  Output: one line `ASSIGN_JSON {...}`."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G   # noqa: E402  (helpers only: workdir with the stand-ins, constants, plant)


def run(inplace: bool, states=None):
    import torch
    from SI_Toolkit.computation_library import PyTorchLibrary
    import Control_Toolkit.Controllers.controller_mpc as cm
    if inplace:
        PyTorchLibrary.assign = staticmethod(lambda v, x: v.copy_(x.detach()))     # torch in-place: views of `v` follow it
    N, H, p, its, steps, resamp = 16, 12, 5, 2, 4, 2                               # the `rpgd_ode_small` fixture's configuration
    cfg = dict(seed=1, mpc_horizon=H, num_rollouts=N, outer_its=its, sample_stdev=0.5, sample_mean=0.0, sample_whole_control_space=True,
               uniform_dist_min=-1.0, uniform_dist_max=1.0, resamp_per=resamp, period_interpolation_inducing_points=p,
               SAMPLING_DISTRIBUTION="uniform", shift_previous=1, warmup=False, warmup_iterations=250, learning_rate=0.05,
               opt_keep_k_ratio=0.25, gradmax_clip=5.0, rtol=1e-3, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-8, mpc_timestep=0.02)
    cm.config_optimizers["rpgd"] = dict(cfg)
    low, high = np.array([-1.0], np.float32), np.array([1.0], np.float32)
    ctrl = cm.controller_mpc("CartPole", (low, high), {})
    ctrl.configure(optimizer_name="rpgd", predictor_specification="ODE")
    opt = ctrl.optimizer
    plant = G.O.Predictor(kind="ODE", dt=0.02, env=G.O.EnvParams(terminal_weight=0.5))
    s = G.initial_state(5)
    out = {}
    for t in range(steps):
        if states is not None:
            s = states[t]
        out[f"s_{t}"] = s.copy()
        u = ctrl.step(s.copy())
        out[f"u_{t}"] = np.asarray(u, np.float32).reshape(-1).copy()
        out[f"u_nom_{t}"] = opt.u_nom.detach().numpy().copy()
        out[f"Q_{t}"] = opt.Q_tf.detach().numpy().copy()
        stp, m_arr, v_arr = opt.opt.get_weights()
        out[f"m_{t}"] = np.asarray(m_arr).copy(); out[f"v_{t}"] = np.asarray(v_arr).copy()
        out[f"ages_{t}"] = opt.trajectory_ages.numpy().copy()
        # the plant follows the VALUE-semantics control sequence in both runs (given from outside), so that each step compares
        # the two semantics on identical optimizer inputs as far as the semantics themselves allow
        s = G.plant_step(plant, s, u)
    return out


def main():
    G.setup_workdir()
    G.inject_constants(G.O.EnvParams(terminal_weight=0.5), 0.02, G.O.mlp_default_weights(0))
    val = run(False)
    steps = len([k for k in val if k.startswith("u_nom_")])
    inp = run(True, states=[val[f"s_{t}"] for t in range(steps)])      # same state sequence: the plant follows the value-semantics run
    fx = np.load(os.path.join(HERE, "golden", "rpgd_ode_small.npz"))
    report = {"steps": steps, "value_run_equals_fixture": bool(all(np.array_equal(val[f"u_{t}"], fx[f"u_{t}"]) and
                                                                  np.array_equal(val[f"Q_{t}"], fx[f"Q_{t}"]) for t in range(steps))),
              "max_abs_diff": {}}
    for t in range(steps):
        for name in ("u", "u_nom", "Q", "m", "v", "ages"):
            k = f"{name}_{t}"
            report["max_abs_diff"][k] = float(np.max(np.abs(val[k].astype(np.float64) - inp[k].astype(np.float64))))
    # what the in-place run returns as u: the first input of ROW best_idx[0] of the warm-started population
    report["inplace_u_is_row_of_new_population"] = [bool(np.any(np.all(np.isclose(inp[f"Q_{t}"][:, 0, :], inp[f"u_{t}"]), axis=-1))) for t in range(steps)]
    print("ASSIGN_JSON " + json.dumps(report))


if __name__ == "__main__":
    main()
