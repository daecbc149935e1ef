"""-m gpu: the GRU predictor (fp32 MFMA path with LDS-resident weights, ctk_gru.h) through the C ABI against
the oracle.  Parity unpinned: the network is build-defined (the reference's GRU lives in the un-vendored
SI_Toolkit; no fixture of it exists), so the oracle's GRU is pinned only by its own definition
(PyTorch gate convention, oracle/ctk_oracle.py:gru_cell); the optimizer logic around it is the golden-pinned one."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from gpu_helpers import apply_env
from test_gpu_mppi import U_TOL

pytestmark = pytest.mark.gpu

# sigmoid/tanh via v_exp/v_rcp (abs err ~2e-7 each) and MFMA summation order, fed back through the
# recurrence for H steps: states to 5e-5 absolute
TRAJ_TOL = dict(rtol=2e-4, atol=5e-5)


def test_gru_plain_rollout_and_hidden_state_match_oracle():
    env = O.EnvParams(terminal_weight=0.4)
    for seed in (0, 3):
        w = O.gru_default_weights(seed)
        pred = O.Predictor("GRU", dt=0.02, env=env, weights=w)
        cost = O.Cost(env)
        e = CtkEngine("mppi", "GRU", num_rollouts=64, mpc_horizon=30, dt=0.02)
        apply_env(e, env)
        with pytest.raises(Exception):
            e.rollout(np.zeros(4, np.float32), np.zeros((3, 30, 1), np.float32))   # weights not set: loud
        with pytest.raises(Exception):
            e.set_predictor_weights(O.mlp_default_weights(0))                        # wrong network: loud
        e.set_predictor_weights(w)
        assert e.predictor_hidden_size() == 64
        np.testing.assert_array_equal(e.predictor_get_hidden(), 0.0)
        rng = np.random.default_rng(seed)
        s = np.array([0.1, 0.2, 1.0, -1.0], np.float32)
        for t in range(3):          # rollouts from a non-trivial carried state after the first update
            Q = rng.uniform(-1, 1, (37, 30, 1)).astype(np.float32)
            traj, J = e.rollout(s, Q, u_prev=0.3)
            to = pred.predict_core(np.tile(s, (37, 1)), Q)
            np.testing.assert_allclose(traj, to, **TRAJ_TOL)
            np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, np.array([0.3], np.float32)), rtol=1e-4, atol=2e-3)
            u = np.float32(rng.uniform(-1, 1))
            pred.update(s, u)
            e.predictor_update(s, [u])
            np.testing.assert_allclose(e.predictor_get_hidden(), pred.hidden, rtol=1e-5, atol=2e-6)
            s = (s + np.float32(0.05) * rng.standard_normal(4)).astype(np.float32)
        # set / reset of the carried state
        hid = rng.uniform(-1, 1, (2, 32)).astype(np.float32)
        e.predictor_set_hidden(hid); pred.hidden = hid.copy()
        Q = rng.uniform(-1, 1, (16, 30, 1)).astype(np.float32)
        np.testing.assert_allclose(e.rollout(s, Q)[0], pred.predict_core(np.tile(s, (16, 1)), Q), **TRAJ_TOL)
        e.predictor_set_hidden(None)
        np.testing.assert_array_equal(e.predictor_get_hidden(), 0.0)
        e.close()


@pytest.mark.parametrize("N,H,p", [(1024, 50, 1), (2048, 40, 10), (16, 5, 2), (70, 12, 5), (8192, 20, 1)])
def test_mppi_gru_matches_oracle(N, H, p):
    """N = 8192 = 512 GRU workgroups takes the unfused path (> CTK_MPPI_FUSE_MAX_BLOCKS_LL): both advance the hidden state."""
    env = O.EnvParams(terminal_weight=0.25)
    w = O.gru_default_weights(1)
    pred = O.Predictor("GRU", dt=0.02, env=env, weights=w)
    o = O.MPPI(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "GRU", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  materialize_trajectories=True)
    apply_env(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(N)
    s = np.array([0.1, -0.2, 2.5, 0.7], np.float32)
    for t in range(3):
        noise = rng.standard_normal((N, o.P, 1)).astype(np.float32)
        uo = o.step(s, noise)
        ug = e.step(s, noise)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, **TRAJ_TOL)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug[0], uo, **U_TOL)
        # optimizer_mppi.py:192: the step advanced the carried state by (s, u_nom[0])
        np.testing.assert_allclose(e.predictor_get_hidden(), pred.hidden, rtol=1e-4, atol=2e-5)
        # keep both sides on the oracle's plan so that later steps compare like with like
        e.set_state(np.concatenate([o.u_nom.reshape(H), np.array([uo], np.float32)]))
        e.predictor_set_hidden(pred.hidden)
        s = (s + np.float32(0.05) * rng.standard_normal(4)).astype(np.float32)
    e.close()


def test_cem_and_random_gru_match_oracle_and_leave_hidden_alone():
    env = O.EnvParams()
    w = O.gru_default_weights(2)
    N, H, K = 500, 20, 50
    hid = np.random.default_rng(9).uniform(-0.5, 0.5, (2, 32)).astype(np.float32)
    s = np.array([0.05, 0.1, 0.4, -0.3], np.float32)
    # CEM
    pred = O.Predictor("GRU", dt=0.02, env=env, weights=w); pred.hidden = hid.copy()
    o = O.CEM(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, cem_outer_it=2, cem_best_k=K)
    e = CtkEngine("cem", "GRU", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=2, cem_best_k=K)
    apply_env(e, env); e.set_predictor_weights(w); e.predictor_set_hidden(hid)
    rng = np.random.default_rng(4)
    for t in range(2):
        eps = rng.standard_normal((2, N, H, 1)).astype(np.float32)
        uo = o.step(s, eps)
        ug = e.step(s, eps)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(e.predictor_get_hidden(), hid)   # no predictor.update in optimizer_cem_tf.py
    e.close()
    # random-action
    pred = O.Predictor("GRU", dt=0.02, env=env, weights=w); pred.hidden = hid.copy()
    o = O.RandomAction(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H)
    e = CtkEngine("random_action", "GRU", num_rollouts=N, mpc_horizon=H, dt=0.02)
    apply_env(e, env); e.set_predictor_weights(w); e.predictor_set_hidden(hid)
    u01 = rng.uniform(0, 1, (N, H, 1)).astype(np.float32)
    uo = o.step(s, u01)
    ug = e.step(s, u01)
    np.testing.assert_allclose(e.read("J"), o.J, rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(ug[0], uo, rtol=1e-6, atol=1e-7)
    e.close()


def test_gru_accepted_by_every_optimizer_and_device_rng_runs():
    # the gradient-based optimizers take the recurrent predictor too (reverse mode: ctk_net.h:NetGru::Bwd; parity: test_gpu_gru_grad.py)
    for opt in ("rpgd", "gradient", "cem_naive_grad", "cem_grad_bharadhwaj"):
        e = CtkEngine(opt, "GRU", num_rollouts=64, mpc_horizon=10, dt=0.02, cem_best_k=8)
        assert "ctk_g_rpgd_descent_split<0, SplitGru>" in e.dominant_kernel()      # N <= 8192: one tile over four waves (ctk_net_split.hip)
        e.set_predictor_weights(O.gru_default_weights(0))
        if opt in ("rpgd", "gradient"):
            e.reset()
        u = e.step(np.array([0.0, 0.0, 0.3, 0.0], np.float32))
        assert np.isfinite(u).all() and abs(u[0]) <= 1.0
        e.close()
    # device sampler path (no host samples): deterministic for a seed, finite, inside the limits
    outs = []
    for rep in range(2):
        e = CtkEngine("mppi", "GRU", num_rollouts=1024, mpc_horizon=50, dt=0.02, seed=11)
        e.set_predictor_weights(O.gru_default_weights(0))
        s = np.array([0.0, 0.0, 0.3, 0.0], np.float32)
        outs.append([float(e.step(s)[0]) for _ in range(5)])
        e.close()
    assert outs[0] == outs[1] and all(np.isfinite(outs[0])) and max(abs(v) for v in outs[0]) <= 1.0
