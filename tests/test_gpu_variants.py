"""-m gpu: SURVEY 8f rank 1 — gradient-tf and cem-naive-grad-tf as thin variants over the same kernels,
against the oracle's restatement of their source text (their modules import tensorflow: parity unpinned
by a reference run)."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from gpu_helpers import apply_env
from test_gpu_rpgd import assert_close_mostly

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pred,N,H,its", [("ODE", 40, 35, 5), ("MLP", 64, 20, 3), ("ODE", 100, 8, 1)])
def test_gradient_matches_oracle(pred, N, H, its):
    env = O.EnvParams(terminal_weight=0.2)
    w = O.mlp_default_weights(4)
    p = O.Predictor(pred, dt=0.02, env=env, weights=w)
    o = O.GradientTF(p, O.Cost(env), num_rollouts=N, mpc_horizon=H, gradient_steps=its, learning_rate=0.05,
                     adam_epsilon=1e-7, gradmax_clip=5.0)
    e = CtkEngine("gradient", pred, num_rollouts=N, mpc_horizon=H, dt=0.02, outer_its=its, learning_rate=0.05,
                  adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-7, gradmax_clip=5.0)
    apply_env(e, env)
    if pred == "MLP":
        e.set_predictor_weights(w)
    rng = np.random.default_rng(N)
    d0 = rng.random((N, H, 1), dtype=np.float32)
    o.optimizer_reset(d0); e.reset(d0)
    np.testing.assert_allclose(e.read("PLAN"), o.Q, rtol=1e-6, atol=1e-7)
    s = np.array([0.05, 0.0, 2.9, 0.3], np.float32)
    for t in range(4):
        assert e.samples_needed() == N
        tail = rng.random((N, 1, 1), dtype=np.float32)
        uo = o.step(s, tail)
        ug = e.step(s, tail)
        tol = dict(rtol=5e-4, atol=5e-4)
        assert_close_mostly(e.read("Q"), o.Q_refined, **tol)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=2e-3, atol=2e-2)
        assert_close_mostly(e.read("PLAN"), o.Q, **tol)
        np.testing.assert_array_equal(e.read("PLAN")[:, -1, 0], o.Q[:, -1, 0])            # the fresh tail, exactly
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, **tol)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-3, atol=1e-3)
        # re-pin (Adam amplifies rounding): continue from the oracle's state
        e.set_state(np.concatenate([o.Q.ravel(), o.opt.m.ravel(), o.opt.v.ravel(), np.zeros(N, np.float32), [float(o.u)],
                                    [o.opt.step_count], [o.count]]).astype(np.float32))
        s = p.step(s.reshape(1, 4), np.array([uo], np.float32))[0] if pred == "ODE" else s
    e.close()


@pytest.mark.parametrize("pred,N,H,K,its", [("ODE", 200, 35, 40, 1), ("ODE", 256, 12, 32, 3), ("MLP", 128, 20, 16, 2)])
def test_cem_naive_grad_matches_oracle(pred, N, H, K, its):
    env = O.EnvParams(terminal_weight=0.2)
    w = O.mlp_default_weights(4)
    p = O.Predictor(pred, dt=0.02, env=env, weights=w)
    o = O.CEMNaiveGrad(p, O.Cost(env), num_rollouts=N, mpc_horizon=H, cem_outer_it=its, cem_best_k=K,
                       cem_initial_action_stdev=0.5, cem_stdev_min=0.1, learning_rate=0.1, gradmax_clip=10.0)
    e = CtkEngine("cem_naive_grad", pred, num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=its, cem_best_k=K,
                  cem_initial_action_stdev=0.5, cem_stdev_min=0.1, learning_rate=0.1, gradmax_clip=10.0)
    apply_env(e, env)
    if pred == "MLP":
        e.set_predictor_weights(w)
    rng = np.random.default_rng(N + H)
    s = np.array([0.02, 0.1, 2.9, -0.5], np.float32)
    for t in range(3):
        noise = rng.standard_normal((its, N, H, 1)).astype(np.float32)
        uo = o.step(s, noise)
        ug = e.step(s, noise)
        # Q moved by lr * clipped gradient: |g| up to gradmax_clip = 10, fp32 gradient error ~1e-5 relative
        # after an H-step reverse sweep => a few 1e-5..1e-4 absolute on the moved samples
        np.testing.assert_allclose(e.read("Q"), o.Q, rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=2e-4, atol=1e-2)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-4, atol=1e-4)
        assert np.all(e.read("STD") <= 10.0) and np.all(e.read("STD") >= np.float32(0.1))
    e.reset(); o.optimizer_reset()
    np.testing.assert_array_equal(e.read("U_NOM"), o.dist_mue)
    e.close()


def test_variant_optimizers_discoverable_and_run():
    from control_toolkit_amd.others.globals_and_utils import import_optimizer_by_name
    from control_toolkit_amd.Predictors import PredictorWrapper
    from control_toolkit_amd.Cost_Functions import CostFunctionWrapper
    from control_toolkit_amd import HipLibrary
    lim = (np.array([-1.0], np.float32), np.array([1.0], np.float32))
    common = dict(predictor=None, cost_function=CostFunctionWrapper(), control_limits=lim, computation_library=HipLibrary(),
                  seed=3, optimizer_logging=True, calculate_optimal_trajectory=False)
    cfgs = {   # the reference's YAML entries (Control_Toolkit_ASF_Template/config_optimizers.yml:22-30, :46-59)
        "gradient-hip": dict(mpc_horizon=35, learning_rate=0.05, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-7, rtol=1e-3,
                             gradient_steps=5, num_rollouts=40, initial_action_stdev=0.5, gradmax_clip=5, warmup=False,
                             warmup_iterations=250),
        "cem-naive-grad-hip": dict(mpc_horizon=35, cem_outer_it=1, num_rollouts=200, cem_stdev_min=0.1,
                                   cem_initial_action_stdev=0.5, cem_best_k=40, learning_rate=0.1, gradmax_clip=10),
    }
    for name, cfg in cfgs.items():
        cls = import_optimizer_by_name(name)
        kw = dict(common); kw["predictor"] = PredictorWrapper()
        opt = cls(**kw, **cfg)
        opt.configure(num_states=4, num_control_inputs=1, dt=0.02, predictor_specification="ODE")
        s = np.array([0.0, 0.0, 0.2, 0.0], np.float32)
        for _ in range(3):
            u = opt.step(s)
            assert np.isfinite(u).all() and abs(float(u)) <= 1.0
        assert opt.logging_values["Q_logged"].shape == (cfg["num_rollouts"], 35, 1)


@pytest.mark.parametrize("pred,N,H,K,its", [("ODE", 32, 50, 8, 2), ("MLP", 64, 16, 16, 3)])
def test_cem_grad_bharadhwaj_matches_oracle(pred, N, H, K, its):
    env = O.EnvParams(terminal_weight=0.2)
    w = O.mlp_default_weights(4)
    p = O.Predictor(pred, dt=0.02, env=env, weights=w)
    kw = dict(cem_outer_it=its, cem_best_k=K, cem_initial_action_stdev=2.0, cem_stdev_min=1e-6, learning_rate=0.05,
              adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-8, gradmax_clip=5.0)
    o = O.CEMGradBharadhwaj(p, O.Cost(env), num_rollouts=N, mpc_horizon=H, **kw)
    e = CtkEngine("cem_grad_bharadhwaj", pred, num_rollouts=N, mpc_horizon=H, dt=0.02, **kw)
    apply_env(e, env)
    if pred == "MLP":
        e.set_predictor_weights(w)
    rng = np.random.default_rng(N * H)
    s = np.array([0.02, 0.1, 2.9, -0.5], np.float32)
    for t in range(3):
        el = rng.standard_normal((K, H, 1)).astype(np.float32)
        rest = rng.standard_normal((its, N - K, H, 1)).astype(np.float32)
        assert e.samples_needed() == el.size + rest.size
        uo = o.step(s, el, rest)
        ug = e.step(s, np.concatenate([el.ravel(), rest.ravel()]))
        tol = dict(rtol=5e-4, atol=5e-4)
        assert_close_mostly(e.read("Q"), o.Q, **tol)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, **tol)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=5e-3, atol=1e-3)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-3, atol=1e-3)
        # re-pin to the oracle's state (mu, std, u, count, m, v, adam step)
        e.set_state(np.concatenate([o.dist_mue.ravel(), o.stdev.ravel(), [float(o.u)], [o.count], o.opt.m.ravel(), o.opt.v.ravel(),
                                    [o.opt.step_count]]).astype(np.float32))
    e.close()
