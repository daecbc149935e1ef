"""-m gpu: the drop-in boundary end to end — controller_mpc.step() with `optimizer: <name>-hip`
and `computation_library: hip`, driven exactly like the reference's controller_mpc
(Controllers/controller_mpc.py:24-109), against the golden closed loop recorded from it."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from helpers import load, env_from, rpgd_kwargs_from
from control_toolkit_amd.Controllers.controller_mpc import controller_mpc
from control_toolkit_amd.Predictors import PredictorWrapper
from control_toolkit_amd.Cost_Functions import CostFunctionWrapper
from test_gpu_mppi import U_TOL
from margins import close

pytestmark = pytest.mark.gpu

LIMITS = (np.array([-1.0], np.float32), np.array([1.0], np.float32))
CTRL_CFG = {"mpc": {"optimizer": "mppi-hip", "predictor_specification": "ODE", "cost_function_specification": "default",
                    "computation_library": "hip", "controller_logging": True, "calculate_optimal_trajectory": True,
                    "device": "gpu:0"}}


class ReplayRng:
    """host generator that replays recorded raw draws (parity mode of the *_hip optimizers)"""
    on_device = False

    def __init__(self, draws):
        self.draws = list(draws)

    def normal(self, shape, dtype=np.float32):
        d = self.draws.pop(0)
        assert list(d.shape) == list(shape)
        return d

    uniform = normal


def build(d, opt_name, opt_cfg, predictor="ODE"):
    env = env_from(d)
    dyn = {k: getattr(env, k) for k in ("g", "m_cart", "m_pole", "L", "u_max", "M_fric", "J_fric")}
    cost = {k: getattr(env, k) for k in ("dd_weight", "ep_weight", "ekp_weight", "cc_weight", "ccrc_weight", "R", "x_scale", "terminal_weight")}
    cfg = {"mpc": dict(CTRL_CFG["mpc"], optimizer=opt_name, predictor_specification=predictor)}
    c = controller_mpc("CartPole", LIMITS, {"target_position": env.target_position, "target_equilibrium": env.target_equilibrium},
                       config_controllers=cfg, config_optimizers={opt_name: opt_cfg},
                       predictor=PredictorWrapper(dyn, weights=d["mlp_weights"]), cost_function=CostFunctionWrapper(cost))
    c.configure()
    return c


def test_controller_mpc_mppi_hip_replays_reference_closed_loop():
    d = load("mppi_interp_ode.npz")
    cfg = dict(seed=1, mpc_horizon=int(d["mpc_horizon"]), num_rollouts=int(d["num_rollouts"]), cc_weight=1.0, R=1.0, LBD=100.0,
               NU=1000.0, SQRTRHOINV=0.03, period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]),
               mpc_timestep=0.02, rng_mode="host")
    c = build(d, "mppi-hip", cfg)
    steps = int(d["steps"])
    c.optimizer.rng = ReplayRng([d[f"noise_{t}"] for t in range(steps)])
    for t in range(steps):
        u = c.step(d[f"s_{t}"])
        np.testing.assert_allclose(u, d[f"u_{t}"][0], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(c.optimizer.u_nom, d[f"u_nom_{t}"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(c.optimizer.logging_values["J_logged"], d[f"J_{t}"], rtol=3e-5)
        # optimal trajectory = rollout of the nominal plan (optimizer_mppi.py:199-202,222-223)
        pred = O.Predictor("ODE", dt=0.02, env=env_from(d))
        ot = pred.predict_core(d[f"s_{t}"].reshape(1, 4), c.optimizer.u_nom)
        np.testing.assert_allclose(c.optimizer.optimal_trajectory, ot, rtol=1e-4, atol=4e-5)
    out = c.get_outputs()
    N, H = int(d["num_rollouts"]), int(d["mpc_horizon"])
    assert out["Q_logged"].shape == (steps, N, H, 1) and out["J_logged"].shape == (steps, N)
    assert out["rollout_trajectories_logged"].shape == (steps, N, H + 1, 4) and out["s_logged"].shape == (steps, 4)
    c.controller_reset()
    np.testing.assert_array_equal(c.optimizer.u_nom, 0.0)


def test_controller_mpc_rpgd_hip_replays_reference_closed_loop():
    d = load("rpgd_ode_small.npz")
    k = rpgd_kwargs_from(d)
    cfg = dict(seed=1, mpc_horizon=int(d["mpc_horizon"]), num_rollouts=int(d["num_rollouts"]), rtol=1e-3, mpc_timestep=0.02,
               rng_mode="host", **k)
    steps = int(d["steps"])
    draws = [d["reset_draws"]] + [d[f"resample_draws_{t}"] for t in range(steps) if f"resample_draws_{t}" in d.files]
    import control_toolkit_amd.Optimizers.optimizer_rpgd_hip as mod
    orig = mod.template_optimizer.__init__

    def patched(self, *a, **kw):        # the reset inside configure() consumes the first draw: inject before
        orig(self, *a, **kw)
        self.rng = ReplayRng(draws)
    mod.template_optimizer.__init__ = patched
    try:
        c = build(d, "rpgd-hip", cfg)
    finally:
        mod.template_optimizer.__init__ = orig
    # controller_logging stays on (CTRL_CFG): optimizer_rpgd.py:428-433's logging_values and :518-521's optimal trajectory / summed stage
    # cost against what the unmodified reference logged (tests/golden/make_golden.py: record_rpgd_logged_outputs)
    N, H = int(d["num_rollouts"]), int(d["mpc_horizon"])
    ages_before = np.zeros(N, np.float32)
    for t in range(steps):
        u = c.step(d[f"s_{t}"])
        np.testing.assert_allclose(u, d[f"u_{t}"], rtol=3e-4, atol=3e-4)
        np.testing.assert_array_equal(c.optimizer.trajectory_ages, d[f"ages_{t}"])
        lv = c.optimizer.logging_values
        close(f"rpgd_ode_small[controller] step {t}", "Q_logged", lv["Q_logged"], d[f"Q_logged_{t}"], rtol=2e-5, atol=2e-5)
        close(f"rpgd_ode_small[controller] step {t}", "J_logged", lv["J_logged"], d[f"J_logged_{t}"], rtol=1e-5)
        close(f"rpgd_ode_small[controller] step {t}", "traj_logged", lv["rollout_trajectories_logged"], d[f"traj_logged_{t}"], rtol=1e-4, atol=2e-5)
        assert c.optimizer.rollout_trajectories is lv["rollout_trajectories_logged"]
        # the ages as get_action saw them (:432 logs before :456-458 / :514).  The reference's torch run hands out a VIEW there, which a
        # later in-place `+= 1` changes on non-resampling steps (the fixture's ages_logged shows it: step 1 logs the post-step ages); the
        # value at logging time is what is reproduced
        np.testing.assert_array_equal(lv["trajectory_ages_logged"], ages_before)
        if t % int(d["resamp_per"]) == 0:
            np.testing.assert_array_equal(d[f"ages_logged_{t}"], ages_before)
        ages_before = d[f"ages_{t}"]
        close(f"rpgd_ode_small[controller] step {t}", "optimal_trajectory", c.optimizer.optimal_trajectory, d[f"optimal_trajectory_{t}"], rtol=1e-4, atol=2e-5)
        close(f"rpgd_ode_small[controller] step {t}", "summed_stage_cost", c.optimizer.summed_stage_cost, d[f"summed_stage_cost_{t}"], rtol=1e-5)
    out = c.get_outputs()
    assert out["rollout_trajectories_logged"].shape == (steps, N, H + 1, 4) and out["trajectory_ages_logged"].shape == (steps, N)


def test_controller_update_attributes_and_cost_reload_reach_the_kernels():
    d = load("mppi_tiny_ode.npz")
    cfg = dict(seed=3, mpc_horizon=20, num_rollouts=256, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
               period_interpolation_inducing_points=1, mpc_timestep=0.02)
    c = build(d, "mppi-hip", cfg)
    s = np.array([0.0, 0.0, 0.05, 0.0], np.float32)
    c.step(s)
    J0 = c.optimizer.logging_values["J_logged"].copy()
    # per-step attribute (reference Controllers/__init__.py:106-107, controller_mpc.py:103)
    c.optimizer.engine.set_state(np.zeros(21, np.float32)); c.optimizer.u = 0.0
    c.optimizer.engine._lib.ctk_reset(c.optimizer.engine._h, None, 0)
    c2 = build(d, "mppi-hip", cfg)
    c2.step(s, updated_attributes={"target_position": 0.1})
    J1 = c2.optimizer.logging_values["J_logged"]
    dd = 600.0 * (0.1 / 0.198) ** 2     # the distance term every stage now pays (x ~ 0)
    assert 0.5 * dd < np.median(J1 - J0) < 1.5 * dd
    assert c2.optimizer.engine.get_param("target_position") == np.float32(0.1)
    # cost YAML hot reload (cost_function_wrapper.py:71-74): flag -> consumed at the top of step()
    c2.cost_function.set_parameters(dd_weight=0.0)
    assert c2.cost_function.reload_cost_parameters_from_config_flag
    c2.step(s, updated_attributes={"target_position": 0.1})
    assert not c2.cost_function.reload_cost_parameters_from_config_flag
    assert c2.optimizer.engine.get_param("dd_weight") == 0.0


@pytest.mark.parametrize("name,cfg", [
    ("cem-hip", dict(seed=1, mpc_horizon=30, cem_outer_it=3, cem_initial_action_stdev=0.5, num_rollouts=512, cem_stdev_min=0.01,
                     cem_best_k=51, warmup=False, warmup_iterations=250, mpc_timestep=0.02)),
    ("random-action-hip", dict(seed=1, mpc_horizon=10, num_rollouts=32, mpc_timestep=0.02)),
    ("mppi-hip", dict(seed=None, mpc_horizon=35, num_rollouts=3500, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
                      period_interpolation_inducing_points=10, mpc_timestep=0.02)),
])
def test_all_optimizers_balance_the_pole_with_device_rng(name, cfg):
    """closed loop with the on-device Philox sampler (performance mode): starting near upright the
    controller keeps the pole up for 60 steps — a behavioural check of the whole pipeline."""
    d = load("mppi_tiny_ode.npz")
    c = build(d, name, cfg)
    pred = O.Predictor("ODE", dt=0.02, env=env_from(d))
    s = np.array([0.0, 0.0, 0.15, 0.0], np.float32)
    for t in range(60):
        u = np.asarray(c.step(s), np.float32).reshape(-1)[0]
        assert -1.0 <= u <= 1.0
        s = pred.step(s.reshape(1, 4), np.array([u], np.float32))[0]
    if name != "random-action-hip":
        assert abs(s[2]) < 0.3, f"pole fell: angle {s[2]}"


def test_cost_yaml_edit_reaches_the_kernels_within_one_step(tmp_path):
    """SURVEY 8f rank 2, hot reload end to end: edit config_cost_function.yml -> watcher flag -> controller_mpc.step
    consumes it (controller_mpc.py:101) -> ctk_set_param -> the very next rollout is costed with the new weights
    (checked against the oracle with those weights)."""
    from control_toolkit_amd.Cost_Functions import CostFunctionUpdater
    y = tmp_path / "config_cost_function.yml"
    y.write_text("cost_function_name_default: default\nCartPole:\n  default:\n    dd_weight: 600.0\n    ep_weight: 20000.0\n")
    N, H = 128, 15
    opt_cfg = dict(seed=3, mpc_horizon=H, num_rollouts=N, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03,
                   period_interpolation_inducing_points=1, mpc_timestep=0.02, rng_mode="host")
    c = controller_mpc("CartPole", LIMITS, {"target_position": 0.0, "target_equilibrium": 1.0},
                       config_controllers={"mpc": dict(CTRL_CFG["mpc"], calculate_optimal_trajectory=False)},
                       config_optimizers={"mppi-hip": opt_cfg}, predictor=PredictorWrapper(),
                       cost_function=CostFunctionWrapper(config_path=str(y), watch=False))
    c.configure()
    rng = np.random.default_rng(0)
    noise = [rng.standard_normal((N, H, 1)).astype(np.float32) for _ in range(2)]
    c.optimizer.rng = ReplayRng(noise)
    s = np.array([0.05, 0.0, 0.4, 0.1], np.float32)

    def oracle_J(env, noise_t, u_nom, u_prev):
        o = O.MPPI(O.Predictor("ODE", dt=0.02, env=env), O.Cost(env), num_rollouts=N, mpc_horizon=H,
                   period_interpolation_inducing_points=1)
        o.u_nom = u_nom.copy(); o.u = np.float32(u_prev)
        o.step(s, noise_t)
        return o.J

    J_ref0 = oracle_J(O.EnvParams(), noise[0], np.zeros((1, H, 1), np.float32), 0.0)
    u0 = c.step(s)
    np.testing.assert_allclose(c.optimizer.logging_values["J_logged"], J_ref0, rtol=3e-5)
    u_nom0 = np.asarray(c.optimizer.u_nom, np.float32).reshape(1, H, 1).copy()
    import time; time.sleep(0.01)
    y.write_text("cost_function_name_default: default\nCartPole:\n  default:\n    dd_weight: 50.0\n    ep_weight: 5000.0\n    ekp_weight: 10.0\n")
    assert c.cost_function.cost_function_updater.poll_now()
    c.step(s)                                                             # consumes the flag, uploads, rolls out
    assert c.optimizer.engine.get_param("dd_weight") == 50.0 and c.optimizer.engine.get_param("ekp_weight") == 10.0
    J_ref1 = oracle_J(O.EnvParams(dd_weight=50.0, ep_weight=5000.0, ekp_weight=10.0), noise[1], u_nom0, float(np.asarray(u0).reshape(-1)[0]))
    np.testing.assert_allclose(c.optimizer.logging_values["J_logged"], J_ref1, rtol=3e-5)
    CostFunctionUpdater.stop_all_watchers()


@pytest.mark.parametrize("opt_name", ["mppi-hip", "rpgd-hip", "cem-hip"])
def test_device_resident_log_equals_host_logging(opt_name):
    """`logging_on_device` keeps Q / J / trajectories / ages of every step in an HBM ring and
    get_outputs() fetches runs of steps in one transfer — same arrays as the per-step host copies of the
    reference path (Controllers/__init__.py:159-178), including across a ring wrap (capacity 3, 8 steps)."""
    from control_toolkit_amd.Optimizers import DeviceLogEntry
    N, H, steps = 96, 12, 8
    base = dict(seed=5, mpc_horizon=H, num_rollouts=N, mpc_timestep=0.02)
    if opt_name == "mppi-hip":
        cfg = dict(base, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03, period_interpolation_inducing_points=3)
    elif opt_name == "cem-hip":
        cfg = dict(base, cem_outer_it=2, cem_best_k=12, cem_initial_action_stdev=0.5, cem_stdev_min=0.01, warmup=False, warmup_iterations=1)
    else:
        d = load("rpgd_ode_small.npz")
        cfg = dict(base, rtol=1e-3, **rpgd_kwargs_from(d))
    outs = []
    for on_device in (False, True):
        c = controller_mpc("CartPole", LIMITS, {"target_position": 0.0, "target_equilibrium": 1.0},
                           config_controllers={"mpc": dict(CTRL_CFG["mpc"], optimizer=opt_name, calculate_optimal_trajectory=False)},
                           config_optimizers={opt_name: dict(cfg, logging_on_device=on_device, logging_capacity=3)},
                           predictor=PredictorWrapper(), cost_function=CostFunctionWrapper())
        c.configure()
        s = np.array([0.02, 0.0, 0.3, -0.1], np.float32)
        for t in range(steps):
            u = c.step(s)
            lv = c.optimizer.logging_values
            if on_device:
                assert isinstance(lv["Q_logged"], DeviceLogEntry) and lv["Q_logged"].shape == (N, H, 1)
                if t == 4:     # a handle is usable like the array it stands for
                    np.testing.assert_array_equal(np.asarray(lv["J_logged"]), c.optimizer.engine.read("J"))
                    np.testing.assert_array_equal(lv["Q_logged"].numpy(), c.optimizer.engine.read("Q"))
            s = (s + np.float32(0.01) * np.float32(t + 1) * np.array([1, -1, 2, 0.5], np.float32)).astype(np.float32)
        outs.append(c.get_outputs())
        if on_device:
            e = c.optimizer.engine
            assert e.log_count() == steps
            with pytest.raises(Exception, match="overwritten"):
                e.log_read("J", 0, 1)                    # capacity 3: step 0 is long gone from the ring
            assert e.log_read("J", steps - 3, 3).shape == (3, N)
    host, dev = outs
    for k in ("Q_logged", "J_logged", "rollout_trajectories_logged", "s_logged", "u_logged", "trajectory_ages_logged"):
        if host[k] is None:
            assert dev[k] is None
            continue
        assert dev[k].shape == host[k].shape and dev[k].shape[0] == steps
        np.testing.assert_array_equal(dev[k], host[k])   # device Philox, same seed: bit-identical runs
