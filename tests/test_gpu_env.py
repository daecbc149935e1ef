"""-m gpu: the environment-agnostic template kernels (csrc/ctk_generic.hip, written against csrc/ctk_env.h only).

 * CartPole through the template kernels == CartPole through its hand-tuned kernels (same arithmetic, other schedule);
 * the second environment (Quad2D: 6 states, 2 control inputs) against the oracle for plain rollouts, MPPI, CEM,
   random-action and RPGD — [N,P,C] / [N,H,C] sample tensors, per-input limits, [H,C] plans, [N,H+1,S] trajectories;
 * the reference-recorded Interpolator fixtures with C = 2 (tests/golden/interpolator.npz, others/Interpolator.py:53-106)
   consumed by the device's interpolation.
Tolerances as in test_gpu_mppi.py / test_gpu_rpgd.py (fp32, other summation order / FMA contraction)."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from helpers import load
from test_gpu_mppi import U_TOL, GOLDEN_U_TOL, J_RTOL
from test_gpu_rpgd import assert_close_mostly

from margins import close

pytestmark = pytest.mark.gpu

QLO, QHI = np.array([-1.0, -0.8], np.float32), np.array([1.0, 0.9], np.float32)   # different limits per input on purpose
S0 = np.array([0.3, -0.2, 0.7, 0.1, 0.25, -0.4], np.float32)


def quad_env(**kw):
    return O.Quad2DParams(terminal_weight=0.4, target_x=0.1, **kw)


def apply_params(e: CtkEngine, env):
    for n in env.param_names():
        e.set_param(n, float(getattr(env, n)))


# ---- CartPole: template kernels == tuned kernels ---------------------------------------------------------------------
@pytest.mark.parametrize("opt", ["mppi", "cem", "random_action", "rpgd"])
def test_cartpole_generic_kernels_match_tuned_kernels(opt):
    N, H, p = 192, 24, (6 if opt in ("mppi", "rpgd") else 1)
    kw = dict(num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=5, materialize_trajectories=True)
    if opt == "cem":
        kw.update(cem_outer_it=2, cem_best_k=20)
    if opt == "rpgd":
        kw.update(outer_its=3, resamp_per=2, opt_keep_k=48, sample_whole_control_space=1)
    a, b = CtkEngine(opt, "ODE", **kw), CtkEngine(opt, "ODE", generic_kernels=True, **kw)
    # the analytic predictor's sampling kernels ARE templates over the environment: generic CartPole = the Env<CartPole>
    # instantiation of ctk_mppi_rollout / ctk_affine_rollout (CEM's tuned path additionally fuses its step into one launch)
    if opt == "mppi":
        assert a.dominant_kernel() == b.dominant_kernel() == "ctk_mppi_rollout<0, 0, true, false>"
    elif opt == "random_action":
        assert a.dominant_kernel() == b.dominant_kernel() == "ctk_affine_rollout<0, 0, true>"
    elif opt == "cem":
        assert a.dominant_kernel() == b.dominant_kernel() == "ctk_cem_fused<0, true>"
    else:
        assert "ctk_g_" in b.dominant_kernel() and "ctk_g_" not in a.dominant_kernel()
    if opt == "rpgd":
        a.reset(); b.reset()
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    pred = O.Predictor("ODE")
    for t in range(4):
        ua, ub = a.step(s), b.step(s)                          # device Philox: identical draws by construction
        np.testing.assert_allclose(b.read("J"), a.read("J"), rtol=3e-5)
        if opt != "rpgd":
            np.testing.assert_allclose(b.read("Q"), a.read("Q"), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(b.read("TRAJ"), a.read("TRAJ"), rtol=1e-4, atol=4e-5)
            np.testing.assert_allclose(b.read("U_NOM"), a.read("U_NOM"), **U_TOL)
        else:
            assert_close_mostly(b.read("PLAN"), a.read("PLAN"), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(ub, ua, rtol=2e-4, atol=2e-4)
        b.set_state(a.get_state())                             # continue from identical warm-start state
        s = pred.step(s.reshape(1, 4), np.array([ua[0]], np.float32))[0]
    a.close(); b.close()


def test_generic_mppi_at_throughput_sizes_runs_the_one_wave_template_kernel():
    """N >= 32768: CartPole's own path switches to its streaming kernels, the template path to ctk_g_rollout (one wave per 64
    rollouts, records merged by separate launches) — the only sizes that kernel's MPPI mode still serves"""
    N, H = 32768 + 64, 8
    kw = dict(num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1, seed=9)
    a, b = CtkEngine("mppi", "ODE", **kw), CtkEngine("mppi", "ODE", generic_kernels=True, **kw)
    assert "ctk_g_rollout" in b.dominant_kernel() and "ctk_mppi_rollout_tp" in a.dominant_kernel()
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(2):
        ua, ub = a.step(s), b.step(s)
        np.testing.assert_allclose(b.read("J"), a.read("J"), rtol=3e-5)
        np.testing.assert_allclose(b.read("U_NOM"), a.read("U_NOM"), **U_TOL)
        np.testing.assert_allclose(ub, ua, rtol=2e-4, atol=2e-4)
    a.close(); b.close()


# ---- Quad2D against the oracle ------------------------------------------------------------------------------------------
def test_quad2d_env_info_and_errors():
    e = CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=64, mpc_horizon=10, dt=0.02, action_low=QLO, action_high=QHI)
    assert (e.S, e.C) == (6, 2) and e.param_names == O.QUAD2D_PARAM_NAMES
    assert e.get_param("pos_weight") == 400.0 and e.mppi_partial_size() == 2 + 10 * 2
    with pytest.raises(ValueError):
        e.step(np.zeros(4, np.float32))                        # a CartPole-sized state
    e.close()
    with pytest.raises(ValueError, match="num_states"):
        CtkEngine("mppi", "ODE", environment="Quad2D", num_states=4, num_control_inputs=1, num_rollouts=8, mpc_horizon=5, dt=0.02)
    eg = CtkEngine("mppi", "GRU", environment="Quad2D", num_rollouts=8, mpc_horizon=5, dt=0.02)     # every predictor on every environment
    assert eg.predictor_weight_count() == O.gru_num_weights(8, 6) and "ctk_g_rollout_split<1, SplitGru," in eg.dominant_kernel()
    eg.close()
    em = CtkEngine("mppi", "MLP", environment="Quad2D", num_rollouts=8, mpc_horizon=5, dt=0.02)
    assert em.predictor_weight_count() == O.mlp_num_weights(8, 6)
    with pytest.raises(ValueError, match="expected 1542"):
        em.set_predictor_weights(np.zeros(1380, np.float32))
    em.close()
    with pytest.raises(ValueError):
        CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=8, mpc_horizon=5, dt=0.02, action_low=[-1, -1, -1])


def test_quad2d_plain_rollout_matches_oracle():
    env = quad_env()
    pred, cost = O.Predictor("ODE", env=env), O.Cost(env)
    e = CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=64, mpc_horizon=30, dt=0.02, action_low=QLO, action_high=QHI)
    apply_params(e, env)
    Q = np.random.default_rng(0).uniform(-1, 1, (37, 30, 2)).astype(np.float32)
    up = np.array([0.2, -0.3], np.float32)
    traj, J = e.rollout(S0, Q, u_prev=up)
    to = pred.predict_core(np.tile(S0, (37, 1)), Q)
    np.testing.assert_allclose(traj, to, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, up), rtol=3e-5)
    # two Euler sub-steps per control step
    e2 = CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=64, mpc_horizon=30, dt=0.02, intermediate_steps=2,
                   action_low=QLO, action_high=QHI)
    apply_params(e2, env)
    traj2, _ = e2.rollout(S0, Q, u_prev=up)
    np.testing.assert_allclose(traj2, O.Predictor("ODE", env=env, intermediate_steps=2).predict_core(np.tile(S0, (37, 1)), Q), rtol=1e-4, atol=2e-5)
    e.close(); e2.close()


@pytest.mark.parametrize("N,H,p", [(1024, 50, 1), (300, 35, 10), (70, 7, 3), (1, 1, 1), (130, 41, 10)])
def test_quad2d_mppi_matches_oracle(N, H, p):
    env = quad_env()
    pred = O.Predictor("ODE", env=env)
    o = O.MPPI(pred, O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  materialize_trajectories=True, action_low=QLO, action_high=QHI)
    apply_params(e, env)
    assert e.inducing_points() == o.P and e.samples_needed() == N * o.P * 2
    rng = np.random.default_rng(N + H)
    s = S0.copy()
    for t in range(3):
        noise = rng.standard_normal((N, o.P, 2)).astype(np.float32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("Q"), o.u_run, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=2e-5 * max(1, H // 25))
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **U_TOL)
        s = pred.step(s.reshape(1, 6), np.asarray(uo, np.float32).reshape(1, 2))[0]
    e.close()


def test_quad2d_mppi_device_draws_and_state_roundtrip():
    N, H, p = 256, 20, 5
    env = quad_env()
    o = O.MPPI(O.Predictor("ODE", env=env), O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "ODE", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  seed=0xABCDEF1234, action_low=QLO, action_high=QHI, materialize_trajectories=True)
    apply_params(e, env)
    for call in range(2):   # the in-kernel Philox stream over the flattened [P*C] columns == the oracle's device_noise
        noise = O.device_noise(seed=0xABCDEF1234, stream=0, call=call, first_row=0, rows=N, cols=o.P * 2, kind="normal").reshape(N, o.P, 2)
        uo, ug = o.step(S0, noise), e.step(S0)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=2e-4)      # the normal transform differs by fp32 rounding of log/sin/cos
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-3, atol=2e-4)
    st = e.get_state()
    assert st.size == H * 2 + 2
    np.testing.assert_array_equal(st[:H * 2], e.read("U_NOM").ravel())
    e.set_state(np.concatenate([np.full(H * 2, 0.25, np.float32), [0.1, -0.1]]).astype(np.float32))
    np.testing.assert_array_equal(e.read("U_NOM"), 0.25)
    e.close()


@pytest.mark.parametrize("N,H,K", [(512, 30, 51), (100, 9, 10)])
def test_quad2d_cem_and_random_match_oracle(N, H, K):
    env = quad_env()
    pred = O.Predictor("ODE", env=env)
    o = O.CEM(pred, O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H, cem_outer_it=3, cem_best_k=K)
    e = CtkEngine("cem", "ODE", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=3, cem_best_k=K,
                  action_low=QLO, action_high=QHI, materialize_trajectories=True)
    apply_params(e, env)
    rng = np.random.default_rng(N)
    s = S0.copy()
    for t in range(3):
        noise = rng.standard_normal((3, N, H, 2)).astype(np.float32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        np.testing.assert_array_equal(e.read("BEST_IDX"), o.best_idx)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-6, atol=1e-7)
        s = pred.step(s.reshape(1, 6), np.asarray(uo, np.float32).reshape(1, 2))[0]
    e.close()
    o = O.RandomAction(pred, O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H)
    e = CtkEngine("random_action", "ODE", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, action_low=QLO, action_high=QHI)
    apply_params(e, env)
    for t in range(2):
        u01 = rng.random((N, H, 2), dtype=np.float32)
        uo, ug = o.step(S0, u01), e.step(S0, u01)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        assert int(e.read("BEST_IDX")[0]) == int(o.best_idx)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-6, atol=1e-7)
    e.close()


def rpgd_state(o):
    return np.concatenate([o.Q.ravel(), o.opt.m.ravel(), o.opt.v.ravel(), o.trajectory_ages.ravel(), np.asarray(o.u, np.float32).reshape(-1),
                           [o.opt.step_count], [o.count]]).astype(np.float32)


@pytest.mark.parametrize("N,H,p,its,dist", [(64, 20, 5, 3, "uniform"), (96, 30, 10, 20, "uniform"), (32, 10, 1, 2, "normal")])
def test_quad2d_rpgd_matches_oracle(N, H, p, its, dist):
    env = quad_env()
    pred = O.Predictor("ODE", env=env)
    o = O.RPGD(pred, O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H, outer_its=its, resamp_per=2,
               period_interpolation_inducing_points=p, SAMPLING_DISTRIBUTION=dist, shift_previous=1, opt_keep_k_ratio=0.25)
    e = CtkEngine("rpgd", "ODE", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  outer_its=its, resamp_per=2, shift_previous=1, opt_keep_k=o.k, sampling_distribution=0 if dist == "uniform" else 1,
                  sample_whole_control_space=1, sample_stdev=0.5, sample_mean=0.0, action_low=QLO, action_high=QHI)
    apply_params(e, env)
    rng = np.random.default_rng(N)
    draw = (lambda shape: rng.random(shape, dtype=np.float32)) if dist == "uniform" else (lambda shape: rng.standard_normal(shape).astype(np.float32))
    d0 = draw((N, o.P, 2))
    o.optimizer_reset(d0); e.reset(d0)
    np.testing.assert_allclose(e.read("PLAN"), o.Q, rtol=1e-6, atol=1e-6)
    tol = dict(rtol=1e-3, atol=3e-3) if its >= 20 else dict(rtol=2e-5, atol=2e-5)   # short descents: observed <= 1.3e-6 (profiles/r04_parity_margins.txt)
    s = S0.copy()
    for t in range(3):
        dr = draw((N - o.k, o.P, 2)) if t % 2 == 0 else None
        assert (e.samples_needed() > 0) == (dr is not None)
        uo, ug = o.step(s, dr), e.step(s, dr)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=2e-3, atol=1e-3)
        assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=max(4, o.Q.size // 400), **tol)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=max(4, o.Q.size // 400), **tol)
        np.testing.assert_array_equal(e.read("AGES"), o.trajectory_ages)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **tol)
        e.set_state(rpgd_state(o))
        s = pred.step(s.reshape(1, 6), np.asarray(uo, np.float32).reshape(1, 2))[0]
    e.close()


def test_quad2d_single_gradient_matches_oracle_adjoint():
    """one Adam iteration from zero moments: m = (1 - beta1) * clip_by_norm(dJ/dQ) — isolates the reverse sweep through
    Env<Quad2D>::step_vjp and the cost gradients (optimizer_rpgd.py:329-338)"""
    N, H = 64, 25
    env = quad_env()
    pred, cost = O.Predictor("ODE", env=env), O.Cost(env)
    e = CtkEngine("rpgd", "ODE", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1,
                  outer_its=1, resamp_per=1000, opt_keep_k=16, sampling_distribution=0, sample_whole_control_space=1, gradmax_clip=1e9,
                  action_low=QLO, action_high=QHI)
    apply_params(e, env)
    d0 = np.random.default_rng(3).random((N, H, 2), dtype=np.float32)
    e.reset(d0)
    Q0 = e.read("PLAN")
    up = np.array([0.05, -0.02], np.float32)
    e.set_state(np.concatenate([Q0.ravel(), np.zeros(2 * N * H * 2 + N, np.float32), up, [0], [1]]).astype(np.float32))   # count 1: no resampling
    e.step(S0, None, u_prev=up)
    _, _, g = O.rollout_cost_and_grad(pred, cost, np.tile(S0, (N, 1)), Q0, up)
    m = e.read("ADAM_M")            # after the warm start: shifted by one step, tail zero-filled (:465,:501)
    np.testing.assert_allclose(m[:, :-1, :], 0.1 * g[:, 1:, :], rtol=2e-3, atol=2e-4 * np.abs(g).max())
    e.close()


# ---- reference-recorded Interpolator fixtures with C = 2 --------------------------------------------------------------
@pytest.mark.parametrize("H,p", [(10, 1), (50, 10), (30, 10), (100, 10), (43, 10), (41, 10), (5, 10), (12, 3)])
def test_device_interpolation_reproduces_reference_fixture_two_inputs(H, p):
    """others/Interpolator.py:53-106 executed by the reference for y [7,P,2] (tests/golden/interpolator.npz).  The device
    interpolates RPGD's sampled inducing points the same way (optimizer_rpgd.py:294): with normal sampling, stdev 1, mean 0
    and limits wider than the draws, optimizer_reset() turns the raw draws y into interpolate(y) exactly."""
    d = load("interpolator.npz")
    y, out = d[f"H{H}_p{p}_C2_y"], d[f"H{H}_p{p}_C2_out"]
    e = CtkEngine("rpgd", "ODE", environment="Quad2D", num_rollouts=7, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  outer_its=1, resamp_per=1, opt_keep_k=1, sampling_distribution=1, sample_stdev=1.0, sample_mean=0.0,
                  action_low=[-100.0, -100.0], action_high=[100.0, 100.0])
    assert e.inducing_points() == y.shape[1]
    e.reset(y)
    np.testing.assert_allclose(e.read("PLAN"), out, rtol=1e-6, atol=1e-6)
    e.close()


# ---- the plug-in boundary with a two-input plant -------------------------------------------------------------------------
@pytest.mark.parametrize("name,cfg", [
    ("mppi-hip", dict(seed=2, mpc_horizon=40, num_rollouts=2048, cc_weight=1.0, R=1.0, LBD=10.0, NU=1000.0, SQRTRHOINV=0.05,
                      period_interpolation_inducing_points=5, mpc_timestep=0.02)),
    ("rpgd-hip", dict(seed=2, mpc_horizon=40, num_rollouts=64, outer_its=5, sample_stdev=0.5, sample_mean=0.0, sample_whole_control_space=True,
                      uniform_dist_min=-1.0, uniform_dist_max=1.0, resamp_per=10, period_interpolation_inducing_points=5,
                      SAMPLING_DISTRIBUTION="uniform", shift_previous=1, warmup=False, warmup_iterations=0, learning_rate=0.05,
                      opt_keep_k_ratio=0.25, gradmax_clip=5.0, rtol=1e-3, adam_beta_1=0.9, adam_beta_2=0.999, adam_epsilon=1e-8,
                      mpc_timestep=0.02)),
])
def test_controller_mpc_flies_the_quadrotor_to_its_target(name, cfg):
    """controller_mpc("Quad2D", ...) with `optimizer: <name>` (reference controller_mpc.py:24-106): predictor, cost and
    optimizer resolve the environment by name; closed loop against the oracle plant the vehicle reaches the target."""
    from control_toolkit_amd.Controllers.controller_mpc import controller_mpc
    c = controller_mpc("Quad2D", (np.array([-1.0, -1.0], np.float32), np.array([1.0, 1.0], np.float32)), {"target_x": 0.4, "target_z": 1.2},
                       config_controllers={"mpc": {"optimizer": name, "predictor_specification": "ODE", "computation_library": "hip",
                                                   "controller_logging": False, "calculate_optimal_trajectory": True, "device": "gpu:0"}},
                       config_optimizers={name: cfg})
    c.configure()
    assert c.optimizer.engine.environment == "Quad2D" and c.optimizer.engine.get_param("target_x") == np.float32(0.4)
    pred = O.Predictor("ODE", env=O.Quad2DParams())
    s = np.array([0.0, 0.0, 1.0, 0.0, 0.0, 0.0], np.float32)
    for t in range(150):
        u = np.asarray(c.step(s), np.float32).reshape(-1)
        assert u.shape == (2,) and np.all(np.abs(u) <= 1.0)
        s = pred.step(s.reshape(1, 6), u.reshape(1, 2))[0]
    assert c.optimizer.optimal_trajectory.shape == (1, 41, 6) and c.optimizer.u_nom.shape == (1, 40, 2)
    assert abs(s[0] - 0.4) < 0.12 and abs(s[2] - 1.2) < 0.12 and abs(s[4]) < 0.3, f"did not reach the target: {s}"
    c.step(s, updated_attributes={"target_x": -0.2})          # per-step attribute reaches the kernels (Controllers/__init__.py:106-107)
    assert c.optimizer.engine.get_param("target_x") == np.float32(-0.2)


# ---- reference-recorded fixtures on the second environment (tests/golden/make_golden.py: the UNMODIFIED optimizer_mppi.py /
#      optimizer_rpgd.py driven through controller_mpc with a 6-state, 2-input plant) -------------------------------------------
def quad_engine_from(d, opt, **kw):
    from helpers import env_from
    e = CtkEngine(opt, "ODE", environment="Quad2D", num_rollouts=int(d["num_rollouts"]), mpc_horizon=int(d["mpc_horizon"]), dt=float(d["dt"]),
                  action_low=d["low"], action_high=d["high"], period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]), **kw)
    apply_params(e, env_from(d))
    return e


@pytest.mark.parametrize("materialize", [True, False])
@pytest.mark.parametrize("case", ["quad2d", "quad2d_p1"])
def test_quad2d_mppi_matches_reference_golden(case, materialize):
    d = load(f"mppi_{case}.npz")
    e = quad_engine_from(d, "mppi", materialize_trajectories=materialize, cc_weight=float(d["cc_weight"]), R=float(d["R"]), LBD=float(d["LBD"]),
                         NU=float(d["NU"]), SQRTRHOINV=float(d["SQRTRHOINV"]))
    H = int(d["mpc_horizon"])
    np.testing.assert_array_equal(e.read("U_NOM"), d["u_nom_init"])
    for t in range(int(d["steps"])):
        u = e.step(d[f"s_{t}"], d[f"noise_{t}"], u_prev=d[f"u_prev_{t}"])
        if materialize:
            close(f"mppi_{case}[materialize={materialize}] step {t}", "q", e.read("Q"), d[f"u_run_{t}"], rtol=1e-6, atol=1e-6)
            close(f"mppi_{case}[materialize={materialize}] step {t}", "traj", e.read("TRAJ"), d[f"traj_{t}"], rtol=1e-4, atol=2e-5)
        close(f"mppi_{case}[materialize={materialize}] step {t}", "j", e.read("J"), d[f"J_{t}"], rtol=J_RTOL)
        close(f"mppi_{case}[materialize={materialize}] step {t}", "u_nom", e.read("U_NOM"), d[f"u_nom_{t}"], **GOLDEN_U_TOL)
        close(f"mppi_{case}[materialize={materialize}] step {t}", "u", u, d[f"u_{t}"], **GOLDEN_U_TOL)
        e.set_state(np.concatenate([d[f"u_nom_{t}"].reshape(H * 2), d[f"u_{t}"].reshape(2)]))
    e.close()


@pytest.mark.parametrize("case", ["quad2d", "quad2d_its20"])
def test_quad2d_rpgd_matches_reference_golden(case):
    from helpers import rpgd_kwargs_from
    d = load(f"rpgd_{case}.npz")
    k = rpgd_kwargs_from(d)
    N = int(d["num_rollouts"])
    e = quad_engine_from(d, "rpgd", outer_its=k["outer_its"], resamp_per=k["resamp_per"], shift_previous=k["shift_previous"],
                         opt_keep_k=int(max(int(N * k["opt_keep_k_ratio"]), 1)), sampling_distribution=0 if k["SAMPLING_DISTRIBUTION"] == "uniform" else 1,
                         sample_whole_control_space=int(k["sample_whole_control_space"]), sample_stdev=k["sample_stdev"], sample_mean=k["sample_mean"],
                         sample_min=k["uniform_dist_min"], sample_max=k["uniform_dist_max"], learning_rate=k["learning_rate"],
                         gradmax_clip=k["gradmax_clip"], adam_beta_1=k["adam_beta_1"], adam_beta_2=k["adam_beta_2"], adam_epsilon=k["adam_epsilon"])
    e.reset(d["reset_draws"])
    np.testing.assert_allclose(e.read("PLAN"), d["Q_init"], rtol=1e-6, atol=1e-7)
    tol = dict(rtol=1e-3, atol=3e-3) if k["outer_its"] >= 20 else dict(rtol=2e-5, atol=2e-5)   # short descents: observed <= 1.3e-6 (profiles/r04_parity_margins.txt)
    count = 0
    for t in range(int(d["steps"])):
        key = f"resample_draws_{t}"
        assert (e.samples_needed() > 0) == (key in d.files)
        u = e.step(d[f"s_{t}"], d[key] if key in d.files else None, u_prev=d[f"u_prev_{t}"])
        count += 1
        close(f"rpgd_{case} step {t}", "u_nom", e.read("U_NOM"), d[f"u_nom_{t}"], **tol)
        close(f"rpgd_{case} step {t}", "u", u, d[f"u_{t}"], **tol)
        close(f"rpgd_{case} step {t}", "plan", e.read("PLAN"), d[f"Q_{t}"], **tol)
        close(f"rpgd_{case} step {t}", "adam_m", e.read("ADAM_M"), d[f"m_{t}"], **tol)
        close(f"rpgd_{case} step {t}", "adam_v", e.read("ADAM_V"), d[f"v_{t}"], **tol)
        np.testing.assert_array_equal(e.read("AGES"), d[f"ages_{t}"])
        e.set_state(np.concatenate([d[f"Q_{t}"].ravel(), d[f"m_{t}"].ravel(), d[f"v_{t}"].ravel(), d[f"ages_{t}"].ravel(), d[f"u_{t}"].ravel(),
                                    [int(d[f"adam_step_{t}"])], [count]]).astype(np.float32))
    e.close()


# ---- the MLP predictor ((S+C)-32-32-S tanh network on the fp32 matrix cores) under the template kernels ------------------------
@pytest.mark.parametrize("opt", ["mppi", "cem", "random_action", "rpgd"])
def test_cartpole_generic_mlp_kernels_match_tuned_mlp_kernels(opt):
    N, H, p = 160, 20, (5 if opt in ("mppi", "rpgd") else 1)
    kw = dict(num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=9, materialize_trajectories=True)
    if opt == "cem":
        kw.update(cem_outer_it=2, cem_best_k=16)
    if opt == "rpgd":
        kw.update(outer_its=3, resamp_per=2, opt_keep_k=40, sample_whole_control_space=1)
    w = O.mlp_default_weights(2)
    a, b = CtkEngine(opt, "MLP", **kw), CtkEngine(opt, "MLP", generic_kernels=True, **kw)
    assert "ctk_g_" in b.dominant_kernel() and ("Mlp" in b.dominant_kernel() or "wide_split" in b.dominant_kernel() or "rpgd_persist" in b.dominant_kernel())   # SplitMlp<.> / the wide RPGD forms
    a.set_predictor_weights(w); b.set_predictor_weights(w)
    if opt == "rpgd":
        a.reset(); b.reset()
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(3):
        ua, ub = a.step(s), b.step(s)
        np.testing.assert_allclose(b.read("J"), a.read("J"), rtol=5e-5, atol=1e-3)
        if opt != "rpgd":
            np.testing.assert_allclose(b.read("Q"), a.read("Q"), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(b.read("TRAJ"), a.read("TRAJ"), rtol=1e-4, atol=2e-5)
        else:
            assert_close_mostly(b.read("PLAN"), a.read("PLAN"), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(ub, ua, rtol=2e-4, atol=2e-4)
        b.set_state(a.get_state())
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    a.close(); b.close()


@pytest.mark.parametrize("N,tiles_per_wg", [(8192, 2), (8176, 1), (4096, 1)])
def test_template_mlp_two_tiles_per_workgroup_at_full_chip_sizes(N, tiles_per_wg):
    """More than one 16-trajectory tile per CU: the two-wave MLP rollout puts two tiles in a four-wave workgroup (each wave its own SIMD);
    ragged populations and smaller ones keep one tile.  Results equal the tuned kernel's either way."""
    kw = dict(num_rollouts=N, mpc_horizon=40, dt=0.02, period_interpolation_inducing_points=10, seed=4)
    w = O.mlp_default_weights(2)
    a, b = CtkEngine("mppi", "MLP", **kw), CtkEngine("mppi", "MLP", generic_kernels=True, **kw)
    assert b.dominant_kernel().startswith("ctk_g_rollout_split<0, SplitMlp<false>, 0, false, %d>" % tiles_per_wg), b.dominant_kernel()
    a.set_predictor_weights(w); b.set_predictor_weights(w)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(3):
        ua, ub = a.step(s), b.step(s)
        np.testing.assert_allclose(b.read("J"), a.read("J"), rtol=5e-5, atol=1e-3)
        np.testing.assert_allclose(b.read("U_NOM"), a.read("U_NOM"), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(ub, ua, rtol=2e-4, atol=2e-4)
        b.set_state(a.get_state())
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    a.close(); b.close()


def test_wide_rpgd_with_its_own_jacobian_launch_also_matches():
    """The template's wide RPGD descent hands every forward step to Jacobian workgroups inside the phase launch; CTK_RPGD_NO_OVERLAP (read once
    per process) keeps the Jacobians in their own launch after each phase launch.  Same tests, child process."""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__),
                        "-k", "(generic_mlp_kernels_match_tuned and rpgd) or quad2d_mlp_gradient_and_rpgd"],
                       env=dict(os.environ, CTK_RPGD_NO_OVERLAP="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def quad_mlp(seed=3):
    env = quad_env()
    w = O.mlp_default_weights(seed, 8, 6)
    return env, w, O.Predictor("MLP", env=env, weights=w)


def test_quad2d_mlp_plain_rollout_and_mppi_match_oracle():
    env, w, pred = quad_mlp()
    cost = O.Cost(env)
    e = CtkEngine("mppi", "MLP", environment="Quad2D", num_rollouts=200, mpc_horizon=30, dt=0.02, period_interpolation_inducing_points=5,
                  action_low=QLO, action_high=QHI, materialize_trajectories=True)
    apply_params(e, env); e.set_predictor_weights(w)
    Q = np.random.default_rng(0).uniform(-1, 1, (37, 30, 2)).astype(np.float32)
    up = np.array([0.2, -0.3], np.float32)
    traj, J = e.rollout(S0, Q, u_prev=up)
    to = pred.predict_core(np.tile(S0, (37, 1)), Q)
    np.testing.assert_allclose(traj, to, rtol=1e-4, atol=3e-5)        # tanh via v_exp/v_rcp, MFMA summation order
    np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, up), rtol=5e-5, atol=1e-3)
    o = O.MPPI(pred, cost, QLO, QHI, num_rollouts=200, mpc_horizon=30, period_interpolation_inducing_points=5)
    rng = np.random.default_rng(1)
    s = S0.copy()
    for t in range(3):
        noise = rng.standard_normal((200, o.P, 2)).astype(np.float32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("Q"), o.u_run, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=5e-5, atol=1e-3)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **U_TOL)
        s = pred.step(s.reshape(1, 6), np.asarray(uo, np.float32).reshape(1, 2))[0]
    e.close()


def test_quad2d_mlp_cem_matches_oracle():
    env, w, pred = quad_mlp(4)
    N, H, K = 256, 20, 25
    o = O.CEM(pred, O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H, cem_outer_it=2, cem_best_k=K)
    e = CtkEngine("cem", "MLP", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=2, cem_best_k=K,
                  action_low=QLO, action_high=QHI)
    apply_params(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(2)
    for t in range(2):
        noise = rng.standard_normal((2, N, H, 2)).astype(np.float32)
        uo, ug = o.step(S0, noise), e.step(S0, noise)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=5e-5, atol=1e-3)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-5, atol=1e-6)
    e.close()


def test_quad2d_mlp_gradient_and_rpgd_match_oracle():
    env, w, pred = quad_mlp(5)
    cost = O.Cost(env)
    N, H = 64, 20
    # one Adam iteration from zero moments: m = (1 - beta1) * dJ/dQ — isolates the MFMA reverse sweep (mlp_step_vjp2) for S = 6, C = 2
    e = CtkEngine("rpgd", "MLP", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1,
                  outer_its=1, resamp_per=1000, opt_keep_k=16, sampling_distribution=0, sample_whole_control_space=1, gradmax_clip=1e9,
                  action_low=QLO, action_high=QHI)
    apply_params(e, env); e.set_predictor_weights(w)
    e.reset(np.random.default_rng(3).random((N, H, 2), dtype=np.float32))
    Q0 = e.read("PLAN")
    up = np.array([0.05, -0.02], np.float32)
    e.set_state(np.concatenate([Q0.ravel(), np.zeros(2 * N * H * 2 + N, np.float32), up, [0], [1]]).astype(np.float32))
    e.step(S0, None, u_prev=up)
    _, _, g = O.rollout_cost_and_grad(pred, cost, np.tile(S0, (N, 1)), Q0, up)
    np.testing.assert_allclose(e.read("ADAM_M")[:, :-1, :], 0.1 * g[:, 1:, :], rtol=2e-3, atol=2e-4 * np.abs(g).max())
    e.close()
    # the full step, several iterations
    o = O.RPGD(pred, cost, QLO, QHI, num_rollouts=N, mpc_horizon=H, outer_its=4, resamp_per=2, period_interpolation_inducing_points=5,
               SAMPLING_DISTRIBUTION="uniform", opt_keep_k_ratio=0.25)
    e = CtkEngine("rpgd", "MLP", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=5,
                  outer_its=4, resamp_per=2, opt_keep_k=o.k, sampling_distribution=0, sample_whole_control_space=1, action_low=QLO, action_high=QHI)
    apply_params(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(6)
    d0 = rng.random((N, o.P, 2), dtype=np.float32)
    o.optimizer_reset(d0); e.reset(d0)
    tol = dict(rtol=2e-4, atol=3e-4)
    s = S0.copy()
    for t in range(3):
        dr = rng.random((N - o.k, o.P, 2), dtype=np.float32) if t % 2 == 0 else None
        uo, ug = o.step(s, dr), e.step(s, dr)
        assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=max(4, o.Q.size // 400), **tol)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=max(4, o.Q.size // 400), **tol)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **tol)
        e.set_state(rpgd_state(o))
        s = pred.step(s.reshape(1, 6), np.asarray(uo, np.float32).reshape(1, 2))[0]
    e.close()


@pytest.mark.parametrize("N,H", [(40, 1), (50, 2), (37, 3), (200, 7), (72, 26), (1100, 12)])
def test_quad2d_mlp_wide_rpgd_edge_shapes_match_oracle(N, H):
    """The wide RPGD descent of the template (Jacobian workgroups inside the phase launch: flags lag the forward pass by four steps; the
    adjoint chain walks whole blocks of its ring depth) at horizons shorter than the lag and the ring, ragged populations, and more tiles
    (N = 1100: 69) than the in-launch form takes (64: there the Jacobians keep their own launch).  Up to 32 tiles these shapes now run the
    one-launch form (ctk_g_rpgd_persist); test_template_rpgd_phase_launches_keep_their_edge_shapes runs them on the phase launches."""
    env, w, pred = quad_mlp()
    cost = O.Cost(env)
    p = 1 if H < 4 else 3
    o = O.RPGD(pred, cost, QLO, QHI, num_rollouts=N, mpc_horizon=H, outer_its=3, resamp_per=2, period_interpolation_inducing_points=p,
               SAMPLING_DISTRIBUTION="uniform", opt_keep_k_ratio=0.25)
    e = CtkEngine("rpgd", "MLP", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  outer_its=3, resamp_per=2, opt_keep_k=o.k, sampling_distribution=0, sample_whole_control_space=1, action_low=QLO, action_high=QHI)
    import os
    want = "wide_split" if (N > 512 or os.environ.get("CTK_RPGD_NO_PERSISTENT")) else "rpgd_persist"      # up to 32 tiles: the one-launch form (round 4)
    assert want in e.dominant_kernel(), e.dominant_kernel()
    apply_params(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(N + H)
    d0 = rng.random((N, o.P, 2), dtype=np.float32)
    o.optimizer_reset(d0); e.reset(d0)
    tol = dict(rtol=2e-4, atol=3e-4)
    s = S0.copy()
    for t in range(2):
        dr = rng.random((N - o.k, o.P, 2), dtype=np.float32) if t % 2 == 0 else None
        uo, ug = o.step(s, dr), e.step(s, dr)
        assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=max(4, o.Q.size // 400), **tol)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=max(4, o.Q.size // 400), **tol)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **tol)
        e.set_state(rpgd_state(o))
        s = pred.step(s.reshape(1, 6), np.asarray(uo, np.float32).reshape(1, 2))[0]
    e.close()


# ---- SURVEY 8e with the second environment: shards of N/2 + the record exchange == one handle of N ----------------
@pytest.mark.parametrize("opt", ["mppi", "cem", "random_action", "rpgd"])
def test_quad2d_two_shards_equal_one_handle(opt):
    two_shards_equal_one_handle(opt, "Quad2D", quad_env(), QLO, QHI, S0)


def two_shards_equal_one_handle(opt, envname, env, lo, hi, s0, kind="ODE", weights=None):
    """records carry [.., H*C] plans: MPPI partials (2 + P*C), best-K candidates (2 + H*C), RPGD keepers (3 + 3*H*C)"""
    import torch
    S, C = env.S, env.C
    N, H, p, K, its = 256, 20, 5, 32, 3
    kw = dict(environment=envname, mpc_horizon=H, dt=0.02, action_low=lo, action_high=hi)
    if opt == "mppi":
        kw.update(period_interpolation_inducing_points=p)
    elif opt == "cem":
        kw.update(cem_outer_it=its, cem_best_k=K)
    elif opt == "rpgd":
        kw.update(period_interpolation_inducing_points=p, outer_its=its, resamp_per=2, shift_previous=1, opt_keep_k=K,
                  sampling_distribution=0, sample_whole_control_space=1, learning_rate=0.05, gradmax_clip=5.0)
    full = CtkEngine(opt, kind, num_rollouts=N, **kw)
    sh = [CtkEngine(opt, kind, num_rollouts=N // 2, global_rollout_offset=i * N // 2, **kw) for i in range(2)]
    for e in sh + [full]:
        apply_params(e, env)
        if weights is not None:
            e.set_predictor_weights(weights)
    pred = O.Predictor(kind, env=env, weights=weights)
    P = O.num_inducing_points(H, p)
    rng = np.random.default_rng(17)
    half = lambda a, i: a[i * N // 2:(i + 1) * N // 2]
    if opt == "mppi":
        rec = full.mppi_partial_size(); assert rec == 2 + P * C
    elif opt == "rpgd":
        rec = sh[0].rpgd_keepers_size(); assert rec == K * (3 + 3 * H * C)
        d0 = rng.random((N, P, C), dtype=np.float32)
        full.reset(d0)
        for i, e in enumerate(sh):
            e.reset(half(d0, i))
    else:
        rec = sh[0].shard_candidates_size(); assert rec == (K if opt == "cem" else 1) * (2 + H * C)
    buf = torch.zeros(2 * rec, dtype=torch.float32, device="cuda")
    s = s0.copy()
    for t in range(4):
        if opt == "mppi":
            noise = rng.standard_normal((N, P, C)).astype(np.float32)
            u_full = full.step(s, noise)
            for i, e in enumerate(sh):
                e.mppi_step_begin(s, buf.data_ptr() + 4 * i * rec, half(noise, i))
            torch.cuda.synchronize()
            us = [e.mppi_step_end(buf.data_ptr(), 2) for e in sh]
        elif opt == "rpgd":
            fresh = [e.rpgd_fresh_rows(2) for e in sh]
            assert fresh == ([N // 2, N // 2 - K] if t % 2 == 0 else [0, 0])
            dr = rng.random((N - K, P, C), dtype=np.float32) if t % 2 == 0 else None
            u_full = full.step(s, dr)
            for i, e in enumerate(sh):
                e.rpgd_step_begin(s, buf.data_ptr() + 4 * i * rec)
            torch.cuda.synchronize()
            us = [e.rpgd_step_end(buf.data_ptr(), 2, None if dr is None else dr[i * N // 2: i * N // 2 + fresh[i]]) for i, e in enumerate(sh)]
            for name in ("PLAN", "ADAM_M", "ADAM_V", "AGES"):
                np.testing.assert_allclose(np.concatenate([e.read(name) for e in sh]), full.read(name), rtol=1e-6, atol=1e-7, err_msg=name)
        else:
            n_it = sh[0].shard_iterations()
            draws = (rng.standard_normal((n_it, N, H, C)) if opt == "cem" else rng.random((n_it, N, H, C))).astype(np.float32)
            u_full = full.step(s, draws if opt == "cem" else draws[0])
            for it in range(n_it):
                for i, e in enumerate(sh):
                    e.shard_iter_begin(s, buf.data_ptr() + 4 * i * rec, half(draws[it], i))
                torch.cuda.synchronize()
                for e in sh:
                    e.shard_iter_end(buf.data_ptr(), 2)
            us = [e.shard_finish() for e in sh]
        assert us[0].shape == (C,)
        np.testing.assert_array_equal(us[0], us[1])
        np.testing.assert_allclose(us[0], u_full, **U_TOL)
        if opt in ("mppi", "cem"):
            np.testing.assert_allclose(sh[0].read("U_NOM"), full.read("U_NOM"), **U_TOL)
        s = pred.step(s.reshape(1, S), u_full.reshape(1, C))[0]
    for e in sh + [full]:
        e.close()


def test_template_rpgd_phase_launches_keep_their_edge_shapes():
    """the phase-launch form of the template's wide RPGD descent (what populations beyond 32 tiles and horizons beyond 64 run) on the edge shapes
    above, in a child process: CTK_RPGD_NO_PERSISTENT is read once per process"""
    import os, subprocess, sys
    env = dict(os.environ, CTK_RPGD_NO_PERSISTENT="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(here, "test_gpu_env.py"),
                        "-k", "edge_shapes and not phase_launches"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
