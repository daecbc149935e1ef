"""-m gpu: the third environment (Hover: 7 states, 3 control inputs — 10 network inputs, i.e. a THIRD layer-1 k-step of the MFMA MLP).

 * the reference-recorded fixtures tests/golden/{mppi,rpgd}_hover_{ode,mlp}.npz (tests/golden/make_golden.py: the UNMODIFIED
   optimizer_mppi.py / optimizer_rpgd.py driven through controller_mpc with a 7-state, 3-input plant);
 * plain rollouts, CEM, random-action and the single-gradient check against the oracle (parity pinned only through the oracle's own
   fixtures above: the reference has no CEM fixture on this plant);
 * the GRU predictor with ten network inputs (ctk_net.h: NetGruT<true>, a third layer-1 k-step; round 4 — until then refused at create):
   the same tests against the oracle, the hidden state carried over closed-loop steps, back-propagation through time with the
   adjoints of network inputs 8 and 9.
Tolerances as in test_gpu_env.py; the GRU's as in test_gpu_gru_grad.py (sigmoid / tanh through v_exp_f32 / v_rcp_f32)."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from helpers import load, env_from, rpgd_kwargs_from, MPPI_HOVER_CASES, RPGD_HOVER_CASES
from test_gpu_mppi import U_TOL, GOLDEN_U_TOL, J_RTOL
from test_gpu_rpgd import assert_close_mostly
from test_gpu_env import apply_params, two_shards_equal_one_handle

from margins import close

pytestmark = pytest.mark.gpu

HLO, HHI = np.array([-1.0, -0.7, -0.5], np.float32), np.array([0.9, 1.0, 0.5], np.float32)
S0 = np.array([0.2, -0.1, -0.3, 0.15, 0.4, -0.2, 0.5], np.float32)
KINDS = ["ODE", "MLP", "GRU"]


def hover_weights(kind, seed):
    return O.mlp_default_weights(seed, 10, 7) if kind == "MLP" else O.gru_default_weights(seed, 10, 7) if kind == "GRU" else None


def j_tol(kind):
    return dict(rtol=1e-4, atol=2e-3) if kind == "GRU" else dict(rtol=5e-5, atol=1e-3) if kind == "MLP" else dict(rtol=3e-5)


def traj_tol(kind, H=25):
    return dict(rtol=2e-4, atol=5e-5 * max(1, H // 25)) if kind == "GRU" else dict(rtol=1e-4, atol=3e-5 * max(1, H // 25))


def hover_env(**kw):
    return O.HoverParams(**kw)


def hover_engine_from(d, opt, **kw):
    pred = str(d["predictor"])
    e = CtkEngine(opt, pred, environment="Hover", num_rollouts=int(d["num_rollouts"]), mpc_horizon=int(d["mpc_horizon"]), dt=float(d["dt"]),
                  action_low=d["low"], action_high=d["high"], period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]), **kw)
    apply_params(e, env_from(d))
    if pred == "MLP":
        e.set_predictor_weights(d["mlp_weights"])
    return e


def test_hover_env_info_and_errors():
    e = CtkEngine("mppi", "ODE", environment="Hover", num_rollouts=8, mpc_horizon=5, dt=0.02)
    assert (e.S, e.C) == (7, 3) and e.param_names == O.HOVER_PARAM_NAMES
    env = hover_env()
    for n in env.param_names():                    # the library's defaults are the oracle's
        assert e.get_param(n) == np.float32(getattr(env, n)), n
    assert "<2," in e.dominant_kernel(), e.dominant_kernel()      # the environment id is the first template argument
    e.close()
    em = CtkEngine("mppi", "MLP", environment="Hover", num_rollouts=8, mpc_horizon=5, dt=0.02)
    import os
    form = "NetMlpT<true>" if os.environ.get("CTK_NET_ONE_WAVE") else "SplitMlp<true>"      # (the child process of test_gpu_gru_grad.py's last test)
    assert em.predictor_weight_count() == O.mlp_num_weights(10, 7) and form in em.dominant_kernel()
    with pytest.raises(ValueError, match="expected"):
        em.set_predictor_weights(np.zeros(O.mlp_num_weights(8, 6), np.float32))
    em.close()
    eg = CtkEngine("mppi", "GRU", environment="Hover", num_rollouts=8, mpc_horizon=5, dt=0.02)     # ten network inputs: the one-wave GRU with a third k-step
    import os as _os
    assert eg.predictor_weight_count() == O.gru_num_weights(10, 7) and ("NetGruT<true>" if _os.environ.get("CTK_NET_ONE_WAVE") else "SplitGru") in eg.dominant_kernel(), eg.dominant_kernel()
    eg.close()
    with pytest.raises(ValueError):
        CtkEngine("mppi", "ODE", environment="Hover", num_rollouts=8, mpc_horizon=5, dt=0.02, action_low=[-1, -1])


@pytest.mark.parametrize("kind", KINDS)
def test_hover_plain_rollout_matches_oracle(kind):
    env = hover_env(target_x=0.3, target_y=-0.2)
    w = hover_weights(kind, 11)
    pred, cost = O.Predictor(kind, env=env, weights=w), O.Cost(env)
    e = CtkEngine("mppi", kind, environment="Hover", num_rollouts=64, mpc_horizon=30, dt=0.02, action_low=HLO, action_high=HHI)
    apply_params(e, env)
    if w is not None:
        e.set_predictor_weights(w)
    Q = np.random.default_rng(0).uniform(-1, 1, (37, 30, 3)).astype(np.float32)
    up = np.array([0.2, -0.3, 0.1], np.float32)
    if kind == "GRU":
        pred.hidden = (0.2 * np.random.default_rng(5).standard_normal((2, 32))).astype(np.float32)
        e.predictor_set_hidden(pred.hidden)
    traj, J = e.rollout(S0, Q, u_prev=up)
    to = pred.predict_core(np.tile(S0, (37, 1)), Q)
    np.testing.assert_allclose(traj, to, **traj_tol(kind))
    np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, up), **(j_tol(kind) if kind != "ODE" else dict(rtol=5e-5)))
    e.close()


@pytest.mark.parametrize("materialize", [True, False])
@pytest.mark.parametrize("case", MPPI_HOVER_CASES)
def test_hover_mppi_matches_reference_golden(case, materialize):
    d = load(f"mppi_{case}.npz")
    e = hover_engine_from(d, "mppi", materialize_trajectories=materialize, cc_weight=float(d["cc_weight"]), R=float(d["R"]), LBD=float(d["LBD"]),
                          NU=float(d["NU"]), SQRTRHOINV=float(d["SQRTRHOINV"]))
    H, mlp = int(d["mpc_horizon"]), str(d["predictor"]) == "MLP"
    np.testing.assert_array_equal(e.read("U_NOM"), d["u_nom_init"])
    for t in range(int(d["steps"])):
        u = e.step(d[f"s_{t}"], d[f"noise_{t}"], u_prev=d[f"u_prev_{t}"])
        if materialize:
            close(f"mppi_{case}[materialize={materialize}] step {t}", "q", e.read("Q"), d[f"u_run_{t}"], rtol=1e-6, atol=1e-6)
            close(f"mppi_{case}[materialize={materialize}] step {t}", "traj", e.read("TRAJ"), d[f"traj_{t}"], rtol=1e-4, atol=3e-5)
        close(f"mppi_{case}[materialize={materialize}] step {t}", "j", e.read("J"), d[f"J_{t}"], rtol=J_RTOL, atol=1e-3 if mlp else 0)
        close(f"mppi_{case}[materialize={materialize}] step {t}", "u_nom", e.read("U_NOM"), d[f"u_nom_{t}"], **GOLDEN_U_TOL)
        close(f"mppi_{case}[materialize={materialize}] step {t}", "u", u, d[f"u_{t}"], **GOLDEN_U_TOL)
        e.set_state(np.concatenate([d[f"u_nom_{t}"].reshape(H * 3), d[f"u_{t}"].reshape(3)]))
    e.close()


@pytest.mark.parametrize("case", RPGD_HOVER_CASES)
def test_hover_rpgd_matches_reference_golden(case):
    d = load(f"rpgd_{case}.npz")
    k = rpgd_kwargs_from(d)
    N = int(d["num_rollouts"])
    e = hover_engine_from(d, "rpgd", outer_its=k["outer_its"], resamp_per=k["resamp_per"], shift_previous=k["shift_previous"],
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
        n_out = max(4, d[f"Q_{t}"].size // 400)
        close(f"rpgd_{case} step {t}", "u_nom", e.read("U_NOM"), d[f"u_nom_{t}"], **tol)
        close(f"rpgd_{case} step {t}", "u", u, d[f"u_{t}"], **tol)
        assert_close_mostly(e.read("PLAN"), d[f"Q_{t}"], max_outliers=n_out, **tol)
        assert_close_mostly(e.read("ADAM_M"), d[f"m_{t}"], max_outliers=n_out, **tol)
        assert_close_mostly(e.read("ADAM_V"), d[f"v_{t}"], max_outliers=n_out, **tol)
        np.testing.assert_array_equal(e.read("AGES"), d[f"ages_{t}"])
        e.set_state(np.concatenate([d[f"Q_{t}"].ravel(), d[f"m_{t}"].ravel(), d[f"v_{t}"].ravel(), d[f"ages_{t}"].ravel(), d[f"u_{t}"].ravel(),
                                    [int(d[f"adam_step_{t}"])], [count]]).astype(np.float32))
    e.close()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("N,H,p", [(1024, 40, 1), (300, 35, 10), (70, 7, 3), (1, 1, 1)])
def test_hover_mppi_matches_oracle(kind, N, H, p):
    env = hover_env(target_x=0.3)
    w = hover_weights(kind, 12)
    pred = O.Predictor(kind, env=env, weights=w)
    o = O.MPPI(pred, O.Cost(env), HLO, HHI, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", kind, environment="Hover", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  materialize_trajectories=True, action_low=HLO, action_high=HHI)
    apply_params(e, env)
    if w is not None:
        e.set_predictor_weights(w)
    assert e.inducing_points() == o.P and e.samples_needed() == N * o.P * 3
    rng = np.random.default_rng(N + H)
    s = S0.copy()
    jt = j_tol(kind)
    for t in range(3):
        noise = rng.standard_normal((N, o.P, 3)).astype(np.float32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("Q"), o.u_run, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(e.read("J"), o.J, **jt)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, **traj_tol(kind, H))
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **U_TOL)
        if kind == "GRU":       # predictor.update(s, u) after every step (optimizer_mppi.py:195-197): the hidden state the next rollouts start from
            np.testing.assert_allclose(e.predictor_get_hidden(), pred.hidden, rtol=1e-5, atol=2e-6)
            s = (s + np.array([0.01, 0.0, -0.02, 0.01, 0.0, 0.02, -0.01], np.float32)).astype(np.float32)
        else:
            s = pred.step(s.reshape(1, 7), np.asarray(uo, np.float32).reshape(1, 3))[0]
    e.close()


def test_hover_mppi_device_draws():
    N, H, p = 256, 20, 5
    env = hover_env()
    o = O.MPPI(O.Predictor("ODE", env=env), O.Cost(env), HLO, HHI, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "ODE", environment="Hover", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  seed=0xABCDEF1234, action_low=HLO, action_high=HHI)
    apply_params(e, env)
    for call in range(2):
        noise = O.device_noise(seed=0xABCDEF1234, stream=0, call=call, first_row=0, rows=N, cols=o.P * 3, kind="normal").reshape(N, o.P, 3)
        uo, ug = o.step(S0, noise), e.step(S0)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=2e-4)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-3, atol=2e-4)
    e.close()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("N,H,K", [(512, 30, 51), (100, 9, 10)])
def test_hover_cem_and_random_match_oracle(kind, N, H, K):
    env = hover_env(target_y=0.25)
    w = hover_weights(kind, 13)
    pred = O.Predictor(kind, env=env, weights=w)
    jt = j_tol(kind)
    o = O.CEM(pred, O.Cost(env), HLO, HHI, num_rollouts=N, mpc_horizon=H, cem_outer_it=3, cem_best_k=K)
    e = CtkEngine("cem", kind, environment="Hover", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=3, cem_best_k=K,
                  action_low=HLO, action_high=HHI)
    apply_params(e, env)
    if w is not None:
        e.set_predictor_weights(w)
    rng = np.random.default_rng(N)
    for t in range(2):
        noise = rng.standard_normal((3, N, H, 3)).astype(np.float32)
        uo, ug = o.step(S0, noise), e.step(S0, noise)
        np.testing.assert_allclose(e.read("J"), o.J, **jt)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-5, atol=1e-6)
    e.close()
    o = O.RandomAction(pred, O.Cost(env), HLO, HHI, num_rollouts=N, mpc_horizon=H)
    e = CtkEngine("random_action", kind, environment="Hover", num_rollouts=N, mpc_horizon=H, dt=0.02, action_low=HLO, action_high=HHI)
    apply_params(e, env)
    if w is not None:
        e.set_predictor_weights(w)
    for t in range(2):
        draws = rng.random((N, H, 3), dtype=np.float32)
        uo, ug = o.step(S0, draws), e.step(S0, draws)
        np.testing.assert_allclose(e.read("J"), o.J, **jt)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), rtol=1e-6, atol=1e-7)
    e.close()


@pytest.mark.parametrize("kind", KINDS)
def test_hover_single_gradient_matches_oracle_adjoint(kind):
    """one Adam iteration from zero moments: m = (1 - beta1) * dJ/dQ — isolates the reverse sweep (E::bwd_tape / mlp_step_vjp2 with the third
    k-step's input adjoints) for S = 7, C = 3"""
    env = hover_env(target_x=0.2, target_y=-0.3)
    w = hover_weights(kind, 14)
    pred, cost = O.Predictor(kind, env=env, weights=w), O.Cost(env)
    N, H = 64, 20
    e = CtkEngine("rpgd", kind, environment="Hover", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1,
                  outer_its=1, resamp_per=1000, opt_keep_k=16, sampling_distribution=0, sample_whole_control_space=1, gradmax_clip=1e9,
                  action_low=HLO, action_high=HHI)
    apply_params(e, env)
    if w is not None:
        e.set_predictor_weights(w)
    if kind == "GRU":           # back-propagation through time from a non-zero hidden state (NetGruT<true>::Bwd: inputs 8, 9 are rows 4g + 2 of the input tile)
        pred.hidden = (0.2 * np.random.default_rng(1).standard_normal((2, 32))).astype(np.float32)
        e.predictor_set_hidden(pred.hidden)
    e.reset(np.random.default_rng(3).random((N, H, 3), dtype=np.float32))
    Q0 = e.read("PLAN")
    up = np.array([0.05, -0.02, 0.1], np.float32)
    e.set_state(np.concatenate([Q0.ravel(), np.zeros(2 * N * H * 3 + N, np.float32), up, [0], [1]]).astype(np.float32))
    e.step(S0, None, u_prev=up)
    _, _, g = O.rollout_cost_and_grad(pred, cost, np.tile(S0, (N, 1)), Q0, up)
    np.testing.assert_allclose(e.read("ADAM_M")[:, :-1, :], 0.1 * g[:, 1:, :], rtol=3e-3 if kind == "GRU" else 2e-3, atol=(3e-4 if kind == "GRU" else 2e-4) * np.abs(g).max())
    e.close()


def test_hover_rpgd_with_gru_matches_oracle():
    """closed-loop RPGD through the ten-input GRU (forward with the third k-step, BPTT, Adam, keep-k, resampling) against the oracle"""
    env = hover_env(target_x=0.25)
    w = hover_weights("GRU", 15)
    pred = O.Predictor("GRU", env=env, weights=w)
    N, H, p, its = 32, 20, 5, 3
    o = O.RPGD(pred, O.Cost(env), HLO, HHI, num_rollouts=N, mpc_horizon=H, outer_its=its, resamp_per=2, period_interpolation_inducing_points=p,
               SAMPLING_DISTRIBUTION="uniform", opt_keep_k_ratio=0.25)
    e = CtkEngine("rpgd", "GRU", environment="Hover", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, outer_its=its,
                  resamp_per=2, opt_keep_k=o.k, sampling_distribution=0, sample_whole_control_space=1, action_low=HLO, action_high=HHI)
    apply_params(e, env)
    e.set_predictor_weights(w)
    rng = np.random.default_rng(8)
    d0 = rng.random((N, o.P, 3), dtype=np.float32)
    o.optimizer_reset(d0); e.reset(d0)
    tol = dict(rtol=5e-4, atol=5e-4)
    s = S0.copy()
    for t in range(3):
        dr = rng.random((N - o.k, o.P, 3), dtype=np.float32) if t % 2 == 0 else None
        uo, ug = o.step(s, dr), e.step(s, dr)
        assert_close_mostly(e.read("PLAN"), o.Q, max_outliers=max(4, o.Q.size // 400), **tol)
        assert_close_mostly(e.read("ADAM_M"), o.opt.m, max_outliers=max(4, o.Q.size // 400), **tol)
        np.testing.assert_allclose(ug, np.asarray(uo).reshape(-1), **tol)
        s = (s + np.array([0.01, 0.0, -0.02, 0.01, 0.0, 0.02, -0.01], np.float32)).astype(np.float32)
    e.close()


@pytest.mark.parametrize("kind", ["ODE", "MLP"])
@pytest.mark.parametrize("opt", ["mppi", "cem", "random_action", "rpgd"])
def test_hover_two_shards_equal_one_handle(opt, kind):
    """SURVEY 8e on the third environment: the record layouts with C = 3"""
    two_shards_equal_one_handle(opt, "Hover", hover_env(target_x=0.3), HLO, HHI, S0, kind=kind,
                                weights=O.mlp_default_weights(15, 10, 7) if kind == "MLP" else None)
