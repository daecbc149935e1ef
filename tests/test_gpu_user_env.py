"""-m gpu: a USER environment — a plant + cost that is NOT in csrc/ (VERDICT r3 missing 3).  The reference imports the concrete cost class and
the plant model at run time (cost_function_wrapper.py:59-66, controller_mpc.py:43,67-73); here the model is ONE C++ header
(include/ctk_user_env.h; this test's is tests/envs/pendulum_env.h) that control_toolkit_amd/build_env.py compiles at configure time into a
library of its own, every optimizer kernel instantiated for it.  No file under csrc/ is edited.

The parity oracle of the model is its NumPy counterpart below (statement by statement the header's arithmetic), plugged into the
oracle's reference-pinned optimizers (MPPI / CEM / random-action / RPGD of oracle/ctk_oracle.py) by subclassing its Predictor and Cost.
Parity of the MODEL is unpinned by nature (it is this test's own); what the test pins is that the kernels compute the model the header
states, under optimizers that are pinned elsewhere."""
import os
from dataclasses import dataclass

import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from test_gpu_mppi import U_TOL
from test_gpu_rpgd import assert_close_mostly

pytestmark = pytest.mark.gpu
HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "envs", "pendulum_env.h")
f32 = np.float32


@dataclass
class PendulumParams:
    g: float = 9.81
    length: float = 0.5
    damping: float = 0.1
    torque_gain: float = 12.0
    target_angle: float = 0.0
    ang_weight: float = 50.0
    vel_weight: float = 0.5
    cc_weight: float = 1.0
    ccrc_weight: float = 2.0
    R: float = 1.0
    terminal_weight: float = 0.0
    S = 2
    C = 1

    def param_names(self):
        return ["g", "length", "damping", "torque_gain", "target_angle", "ang_weight", "vel_weight", "cc_weight", "ccrc_weight", "R", "terminal_weight"]


def pend_k(p, dt, isteps=1):
    """CtkUserEnv::derive of tests/envs/pendulum_env.h (double -> fp32 once)"""
    return dict(dt=f32(dt / isteps), gl=f32(np.float64(f32(p.g)) / np.float64(f32(p.length))), c=f32(p.damping), kU=f32(p.torque_gain),
                target=f32(p.target_angle), ang_w=f32(p.ang_weight), vel_w=f32(p.vel_weight),
                ccR=f32(np.float64(f32(p.cc_weight)) * np.float64(f32(p.R))), ccrc=f32(p.ccrc_weight), tw=f32(p.terminal_weight))


class PendulumPredictor(O.Predictor):
    def _ode_step(self, s, q):
        k = pend_k(self.env, self.dt, self.intermediate_steps)
        th, om = s[:, 0].copy(), s[:, 1].copy()
        u = self._q2(q)[:, 0]
        for _ in range(self.intermediate_steps):
            al = k["gl"] * np.sin(th) - k["c"] * om + k["kU"] * u
            th, om = (th + k["dt"] * om).astype(f32), (om + k["dt"] * al).astype(f32)
        return np.stack([th, om], 1).astype(f32)

    def _ode_vjp(self, s, q, lam):
        k = pend_k(self.env, self.dt, 1)
        a_al = k["dt"] * lam[:, 1]
        ds = np.stack([lam[:, 0] + a_al * k["gl"] * np.cos(s[:, 0]), lam[:, 1] + k["dt"] * lam[:, 0] - k["c"] * a_al], 1)
        return ds.astype(f32), (k["kU"] * a_al)[:, None].astype(f32)


class PendulumCost(O.Cost):
    def _k(self):
        return pend_k(self.env, self.dt, 1)       # "ccR" is what oracle.input_cost_grad reads; env.ccrc_weight likewise

    def _state(self, states):
        k = self._k()
        return (k["ang_w"] * (f32(1.0) - np.cos(states[..., 0] - k["target"]))).astype(f32)

    def _get_stage_cost(self, states, inputs, previous_input):
        k = self._k()
        u = inputs[..., 0]
        prev = self._prev_inputs(inputs, previous_input)[..., 0]
        d = u - prev
        return (self._state(states) + k["vel_w"] * states[..., 1] * states[..., 1] + (k["ccR"] * u * u + k["ccrc"] * d * d)).astype(f32)

    def get_terminal_cost(self, terminal_states):
        return (self._k()["tw"] * self._state(terminal_states)).astype(f32)

    def state_grad(self, states, terminal: bool):
        k = self._k()
        gth = k["ang_w"] * np.sin(states[:, 0] - k["target"])
        if terminal:
            return np.stack([k["tw"] * gth, np.zeros_like(gth)], 1).astype(f32)
        return np.stack([gth, f32(2.0) * k["vel_w"] * states[:, 1]], 1).astype(f32)


@pytest.fixture(scope="module")
def pendulum():
    from control_toolkit_amd.build_env import register_environment
    return register_environment(HEADER)       # compiled here on first use (cached by content under control_toolkit_amd/_env_builds/)


def engine(name, opt, env, **kw):
    e = CtkEngine(opt, "ODE", environment=name, dt=0.02, **kw)
    for n in env.param_names():
        e.set_param(n, float(getattr(env, n)))
    return e


def test_user_environment_is_described_by_its_library(pendulum):
    from control_toolkit_amd import _capi
    assert pendulum == "Pendulum"
    S, C, names = _capi.environment_info("Pendulum")
    assert (S, C) == (2, 1) and list(names) == PendulumParams().param_names()
    d = _capi.environment_defaults("Pendulum")
    for n in names:
        assert d[n] == f32(getattr(PendulumParams(), n)), n
    e = CtkEngine("mppi", "ODE", environment="Pendulum", num_rollouts=64, mpc_horizon=10, dt=0.02)
    assert (e.S, e.C) == (2, 1) and "<3," in e.dominant_kernel(), e.dominant_kernel()      # CTK_ENV_USER = 3 is the first template argument
    e.close()
    # the same library still carries the built environments
    lib, _ = _capi.environment_library("Pendulum")
    assert lib.ctk_environment_name(0) == b"CartPole" and lib.ctk_environment_name(3) == b"Pendulum"
    assert _capi.load_library().ctk_environment_name(3) is None                                # ... and the product library has no fourth one


@pytest.mark.parametrize("N,H,p", [(1024, 50, 1), (200, 30, 10), (8192, 20, 5)])
def test_user_env_mppi_matches_oracle(pendulum, N, H, p):
    env = PendulumParams(terminal_weight=0.5, target_angle=0.1)
    pred = PendulumPredictor("ODE", dt=0.02, env=env)
    o = O.MPPI(pred, PendulumCost(env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = engine(pendulum, "mppi", env, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p, materialize_trajectories=True)
    rng = np.random.default_rng(N)
    s = np.array([2.6, -0.4], f32)
    for t in range(3):
        noise = rng.standard_normal((N, o.P, 1)).astype(f32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug[0], uo, **U_TOL)
        s = pred.step(s.reshape(1, 2), np.array([[uo]], f32))[0]
    e.close()


def test_user_env_cem_random_and_plain_rollouts_match_oracle(pendulum):
    env = PendulumParams(terminal_weight=0.3)
    pred, cost = PendulumPredictor("ODE", dt=0.02, env=env), PendulumCost(env)
    N, H, K, its = 512, 25, 50, 3
    o = O.CEM(pred, cost, num_rollouts=N, mpc_horizon=H, cem_outer_it=its, cem_best_k=K)
    e = engine(pendulum, "cem", env, num_rollouts=N, mpc_horizon=H, cem_outer_it=its, cem_best_k=K, materialize_trajectories=True)
    assert e.dominant_kernel().startswith("ctk_cem_fused<3")                # the one-launch CEM step, instantiated for the user model
    rng = np.random.default_rng(5)
    s = np.array([3.0, 0.2], f32)
    for t in range(2):
        noise = rng.standard_normal((its, N, H, 1)).astype(f32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("Q"), o.Q, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-5, atol=2e-6)
    Q = rng.uniform(-1, 1, (33, H, 1)).astype(f32)
    traj, J = e.rollout(s, Q, u_prev=0.25)
    to = pred.predict_core(np.tile(s, (33, 1)), Q)
    np.testing.assert_allclose(traj, to, rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, np.array([0.25], f32)), rtol=3e-5)
    e.close()
    r = O.RandomAction(pred, cost, num_rollouts=320, mpc_horizon=35)
    er = engine(pendulum, "random_action", env, num_rollouts=320, mpc_horizon=35)
    u01 = rng.random((320, 35, 1), dtype=f32)
    np.testing.assert_array_equal(er.step(s, u01)[0], r.step(s, u01))
    er.close()


@pytest.mark.parametrize("N,H,p,its", [(64, 20, 5, 5), (256, 50, 10, 3)])
def test_user_env_rpgd_matches_oracle(pendulum, N, H, p, its):
    """reverse mode through the user model: step_vjp, stage / terminal / input gradients"""
    env = PendulumParams(terminal_weight=0.4)
    pred = PendulumPredictor("ODE", dt=0.02, env=env)
    o = O.RPGD(pred, PendulumCost(env), num_rollouts=N, mpc_horizon=H, outer_its=its, resamp_per=2, period_interpolation_inducing_points=p,
               SAMPLING_DISTRIBUTION="uniform", shift_previous=1, learning_rate=0.05, opt_keep_k_ratio=0.25, gradmax_clip=5.0)
    e = engine(pendulum, "rpgd", env, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p, outer_its=its, resamp_per=2,
               shift_previous=1, opt_keep_k=o.k, sampling_distribution=0, sample_min=-1.0, sample_max=1.0, learning_rate=0.05, gradmax_clip=5.0)
    rng = np.random.default_rng(N + its)
    d0 = rng.random((N, o.P, 1), dtype=f32)
    o.optimizer_reset(d0); e.reset(d0)
    s = np.array([2.9, 0.3], f32)
    for t in range(3):
        dr = rng.random((N - o.k, o.P, 1), dtype=f32) if t % 2 == 0 else None
        uo, ug = o.step(s, dr), e.step(s, dr)
        assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=6, rtol=2e-4, atol=2e-4)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=6, rtol=2e-4, atol=2e-4)
        np.testing.assert_array_equal(e.read("AGES"), o.trajectory_ages)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-3, atol=1e-3)
        e.set_state(np.concatenate([o.Q.ravel(), o.opt.m.ravel(), o.opt.v.ravel(), o.trajectory_ages.ravel(), [float(o.u)], [o.opt.step_count], [o.count]]).astype(f32))
        s = pred.step(s.reshape(1, 2), np.array([[uo]], f32))[0]
    e.close()


def test_user_env_with_a_network_predictor_and_through_controller_mpc(pendulum):
    """network predictors come for free (S + C = 3 inputs, S = 2 outputs), and `environment_name: Pendulum` resolves in the plug-in"""
    env = PendulumParams(terminal_weight=0.2)
    w = O.mlp_default_weights(2, 3, 2)
    pred = PendulumPredictor("MLP", dt=0.02, env=env, weights=w)
    N, H = 256, 20
    o = O.MPPI(pred, PendulumCost(env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=5)
    e = CtkEngine("mppi", "MLP", environment=pendulum, num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=5, materialize_trajectories=True)
    for n in env.param_names():
        e.set_param(n, float(getattr(env, n)))
    e.set_predictor_weights(w)
    rng = np.random.default_rng(3)
    s = np.array([0.4, -0.2], f32)
    noise = rng.standard_normal((N, o.P, 1)).astype(f32)
    uo, ug = o.step(s, noise), e.step(s, noise)
    np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(e.read("J"), o.J, rtol=5e-5, atol=1e-3)
    np.testing.assert_allclose(ug[0], uo, **U_TOL)
    e.close()
    # the plug-in boundary: controller_mpc with environment_name = the user model's NAME, cost weights by name
    from control_toolkit_amd.Controllers.controller_mpc import controller_mpc
    from control_toolkit_amd.Predictors import PredictorWrapper
    from control_toolkit_amd.Cost_Functions import CostFunctionWrapper
    lim = (np.array([-1.0], f32), np.array([1.0], f32))
    cfg = {"mpc": {"optimizer": "mppi-hip", "predictor_specification": "ODE", "cost_function_specification": "default", "computation_library": "hip",
                   "controller_logging": True, "calculate_optimal_trajectory": False, "device": "gpu:0"}}
    ocfg = dict(seed=1, mpc_horizon=30, num_rollouts=512, cc_weight=1.0, R=1.0, LBD=100.0, NU=1000.0, SQRTRHOINV=0.03, period_interpolation_inducing_points=10,
                mpc_timestep=0.02, rng_mode="host")
    c = controller_mpc("Pendulum", lim, {}, config_controllers=cfg, config_optimizers={"mppi-hip": ocfg},
                       predictor=PredictorWrapper(environment_name="Pendulum"),
                       cost_function=CostFunctionWrapper({"ang_weight": 70.0, "terminal_weight": 0.5}, watch=False, environment_name="Pendulum"))
    c.configure()
    assert c.optimizer.engine.environment == "Pendulum" and c.optimizer.engine.get_param("ang_weight") == 70.0
    env2 = PendulumParams(ang_weight=70.0, terminal_weight=0.5)
    pred2 = PendulumPredictor("ODE", dt=0.02, env=env2)
    o2 = O.MPPI(pred2, PendulumCost(env2), num_rollouts=512, mpc_horizon=30, period_interpolation_inducing_points=10)
    from test_gpu_controller import ReplayRng
    draws = [rng.standard_normal((512, o2.P, 1)).astype(f32) for _ in range(3)]
    c.optimizer.rng = ReplayRng([d.copy() for d in draws])
    s = np.array([2.2, 0.1], f32)
    for t in range(3):
        uo = o2.step(s, draws[t])
        u = c.step(s)
        np.testing.assert_allclose(u, uo, **U_TOL)
        np.testing.assert_allclose(c.optimizer.logging_values["J_logged"], o2.J, rtol=3e-5)
        s = pred2.step(s.reshape(1, 2), np.array([[uo]], f32))[0]
