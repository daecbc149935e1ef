"""-m gpu: the recurrent predictor under the template kernels (csrc/ctk_net.h: NetGru — one wave per 16-trajectory tile,
164 MFMAs per step forward, 172 reverse): forward against CartPole's 4-wave GRU kernels and the oracle, reverse mode
(back-propagation through time over the horizon, what autograd does for the reference at optimizer_rpgd.py:329-333) against
the oracle's hand-written adjoint, which tests/test_oracle_golden.py checks against torch autograd in fp64."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from test_gpu_mppi import U_TOL
from test_gpu_rpgd import assert_close_mostly
from test_gpu_env import QLO, QHI, S0, quad_env, apply_params, rpgd_state

pytestmark = pytest.mark.gpu
# GRU tolerances as in test_gpu_gru.py: sigmoid / tanh through v_exp_f32 / v_rcp_f32, MFMA summation order
TRAJ_TOL = dict(rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("opt", ["mppi", "cem", "random_action"])
def test_cartpole_generic_gru_matches_four_wave_gru(opt):
    N, H, p = 96, 20, (5 if opt == "mppi" else 1)
    kw = dict(num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=4, materialize_trajectories=True)
    if opt == "cem":
        kw.update(cem_outer_it=2, cem_best_k=12)
    w = O.gru_default_weights(3)
    a, b = CtkEngine(opt, "GRU", **kw), CtkEngine(opt, "GRU", generic_kernels=True, **kw)
    import os
    form = "NetGru" if os.environ.get("CTK_NET_ONE_WAVE") else "ctk_g_rollout_split<0, SplitGru"        # (the child process of the last test in this file)
    assert form in b.dominant_kernel() and "ctk_g_" not in a.dominant_kernel()             # template (ctk_net_split.hip) vs tuned (ctk_gru.h)
    a.set_predictor_weights(w); b.set_predictor_weights(w)
    h0 = (0.2 * np.random.default_rng(0).standard_normal((2, 32))).astype(np.float32)
    a.predictor_set_hidden(h0); b.predictor_set_hidden(h0)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(3):
        ua, ub = a.step(s), b.step(s)
        np.testing.assert_allclose(b.read("TRAJ"), a.read("TRAJ"), **TRAJ_TOL)
        np.testing.assert_allclose(b.read("J"), a.read("J"), rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(ub, ua, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(b.predictor_get_hidden(), a.predictor_get_hidden(), rtol=1e-5, atol=2e-6)   # MPPI advances it (optimizer_mppi.py:192)
        b.set_state(a.get_state())
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    a.close(); b.close()


def test_quad2d_gru_mppi_matches_oracle_with_hidden_state_carry():
    env = quad_env()
    w = O.gru_default_weights(5, 8, 6)
    pred = O.Predictor("GRU", env=env, weights=w)
    N, H, p = 128, 20, 5
    o = O.MPPI(pred, O.Cost(env), QLO, QHI, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "GRU", environment="Quad2D", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  action_low=QLO, action_high=QHI, materialize_trajectories=True)
    apply_params(e, env)
    assert e.predictor_weight_count() == O.gru_num_weights(8, 6)
    e.set_predictor_weights(w)
    rng = np.random.default_rng(1)
    s = S0.copy()
    for t in range(3):
        noise = rng.standard_normal((N, o.P, 2)).astype(np.float32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, **TRAJ_TOL)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **U_TOL)
        np.testing.assert_allclose(e.predictor_get_hidden(), pred.hidden, rtol=1e-5, atol=2e-6)   # predictor.update(s, u) after every step
        s = s + np.array([0.01, 0.0, -0.02, 0.01, 0.0, 0.02], np.float32)
    e.close()


@pytest.mark.parametrize("envname", ["CartPole", "Quad2D"])
def test_gru_single_gradient_matches_oracle_bptt(envname):
    """one Adam iteration from zero moments: m = (1 - beta1) * dJ/dQ — isolates NetGru::Bwd (gate adjoints + the transposed
    products, hidden-state adjoints carried over the horizon)"""
    env = quad_env() if envname == "Quad2D" else O.EnvParams(terminal_weight=0.3)
    S, C = env.S, env.C
    w = O.gru_default_weights(6, S + C, S)
    pred = O.Predictor("GRU", env=env, weights=w)
    pred.hidden = (0.2 * np.random.default_rng(1).standard_normal((2, 32))).astype(np.float32)
    cost = O.Cost(env)
    N, H = 48, 15
    lim = dict(action_low=QLO, action_high=QHI) if C == 2 else {}
    e = CtkEngine("rpgd", "GRU", environment=envname, num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1,
                  outer_its=1, resamp_per=1000, opt_keep_k=12, sampling_distribution=0, sample_whole_control_space=1, gradmax_clip=1e9, **lim)
    for n in env.param_names():
        e.set_param(n, float(getattr(env, n)))
    e.set_predictor_weights(w)
    e.predictor_set_hidden(pred.hidden)
    e.reset(np.random.default_rng(3).random((N, H, C), dtype=np.float32))
    Q0 = e.read("PLAN")
    s = S0 if C == 2 else np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    up = np.array([0.05, -0.02], np.float32)[:C]
    e.set_state(np.concatenate([Q0.ravel(), np.zeros(2 * N * H * C + N, np.float32), up, [0], [1]]).astype(np.float32))
    e.step(s, None, u_prev=up)
    _, _, g = O.rollout_cost_and_grad(pred, cost, np.tile(s, (N, 1)), Q0, up)
    np.testing.assert_allclose(e.read("ADAM_M")[:, :-1, :], 0.1 * g[:, 1:, :], rtol=3e-3, atol=3e-4 * np.abs(g).max())
    e.close()


def test_cartpole_rpgd_with_gru_matches_oracle():
    env = O.EnvParams(terminal_weight=0.3)
    w = O.gru_default_weights(2)
    pred = O.Predictor("GRU", env=env, weights=w)
    N, H, p, its = 32, 20, 5, 3
    o = O.RPGD(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, outer_its=its, resamp_per=2, period_interpolation_inducing_points=p,
               SAMPLING_DISTRIBUTION="uniform", opt_keep_k_ratio=0.25)
    e = CtkEngine("rpgd", "GRU", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, outer_its=its, resamp_per=2,
                  opt_keep_k=o.k, sampling_distribution=0, sample_whole_control_space=1)
    for n in env.param_names():
        e.set_param(n, float(getattr(env, n)))
    e.set_predictor_weights(w)
    rng = np.random.default_rng(8)
    d0 = rng.random((N, o.P, 1), dtype=np.float32)
    o.optimizer_reset(d0); e.reset(d0)
    tol = dict(rtol=5e-4, atol=5e-4)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(3):
        dr = rng.random((N - o.k, o.P, 1), dtype=np.float32) if t % 2 == 0 else None
        uo, ug = o.step(s, dr), e.step(s, dr)
        assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=max(4, o.Q.size // 400), **tol)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=max(4, o.Q.size // 400), **tol)
        np.testing.assert_allclose(ug[0], uo, **tol)
        e.set_state(np.concatenate([o.Q.ravel(), o.opt.m.ravel(), o.opt.v.ravel(), o.trajectory_ages.ravel(), [float(o.u)], [o.opt.step_count], [o.count]]).astype(np.float32))
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    e.close()


@pytest.mark.parametrize("N,H", [(40, 12), (17, 3), (1, 1), (200, 33)])
def test_cartpole_rpgd_with_gru_ragged_populations_match_oracle(N, H):
    """the four-wave form (ctk_net_split.hip) owns 16 plans per workgroup: populations that are not multiples of 16 (a partial last tile),
    a single plan, a horizon of one"""
    env = O.EnvParams(terminal_weight=0.3)
    w = O.gru_default_weights(3)
    pred = O.Predictor("GRU", env=env, weights=w)
    pred.hidden = (0.1 * np.random.default_rng(2).standard_normal((2, 32))).astype(np.float32)
    its, p = 2, 1
    k = max(N // 4, 1)
    o = O.RPGD(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, outer_its=its, resamp_per=1000, period_interpolation_inducing_points=p,
               SAMPLING_DISTRIBUTION="uniform", opt_keep_k_ratio=k / N)
    e = CtkEngine("rpgd", "GRU", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, outer_its=its, resamp_per=1000,
                  opt_keep_k=o.k, sampling_distribution=0, sample_whole_control_space=1)
    assert "SplitGru" in e.dominant_kernel()
    for n in env.param_names():
        e.set_param(n, float(getattr(env, n)))
    e.set_predictor_weights(w)
    e.predictor_set_hidden(pred.hidden)
    d0 = np.random.default_rng(N).random((N, o.P, 1), dtype=np.float32)
    o.optimizer_reset(d0); e.reset(d0)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    tol = dict(rtol=5e-4, atol=5e-4)
    dr = np.random.default_rng(N + 1).random((N - o.k, o.P, 1), dtype=np.float32)      # the first step resamples (count % resamp_per == 0)
    uo, ug = o.step(s, dr), e.step(s, dr if N > o.k else None)
    assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=max(2, o.Q.size // 400), **tol)
    assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=max(2, o.Q.size // 400), **tol)
    np.testing.assert_allclose(e.read("J"), o.J, rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(ug[0], uo, **tol)
    e.close()


def test_one_wave_network_forms_also_match_oracle():
    """Populations above 8 192 (and horizons whose states do not fit LDS) keep one wave per tile (ctk_generic_net.hip with ctk_net.h:
    NetGru / NetMlpT, forward and reverse); their diagnostic switches (read once per process) put them under the same oracle / golden
    tests in a child process: the GRU tests of this file, the third environment's suite and the second environment's MLP tests."""
    import os, subprocess, sys
    env = dict(os.environ, CTK_RPGD_NET_ONE_WAVE="1", CTK_NET_ONE_WAVE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(here, "test_gpu_gru_grad.py"),
                        os.path.join(here, "test_gpu_hover.py"), os.path.join(here, "test_gpu_env.py"),
                        "-k", "(single_gradient or rpgd_with_gru_matches or quad2d_gru_mppi or generic_gru_matches or hover or mlp) and not one_wave and not shards and not two_tiles and not own_jacobian and not edge_shapes"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
