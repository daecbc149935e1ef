"""-m gpu: the MLP predictor (fp32 MFMA path, ctk_mlp.h) through the C ABI — MPPI against the
reference-recorded golden, MPPI/CEM/random-action/plain rollouts against the oracle."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from helpers import load
from gpu_helpers import mppi_engine_from, apply_env
from test_gpu_mppi import U_TOL, GOLDEN_U_TOL, J_RTOL

from margins import close

pytestmark = pytest.mark.gpu


def test_mlp_plain_rollout_matches_oracle():
    env = O.EnvParams(terminal_weight=0.4)
    for seed in (0, 5):
        w = O.mlp_default_weights(seed)
        pred = O.Predictor("MLP", dt=0.02, env=env, weights=w)
        cost = O.Cost(env)
        e = CtkEngine("mppi", "MLP", num_rollouts=64, mpc_horizon=30, dt=0.02)
        apply_env(e, env)
        with pytest.raises(Exception):
            e.rollout(np.zeros(4, np.float32), np.zeros((3, 30, 1), np.float32))   # weights not set: loud
        e.set_predictor_weights(w)
        Q = np.random.default_rng(seed).uniform(-1, 1, (37, 30, 1)).astype(np.float32)
        s = np.array([0.1, 0.2, 1.0, -1.0], np.float32)
        traj, J = e.rollout(s, Q, u_prev=0.3)
        to = pred.predict_core(np.tile(s, (37, 1)), Q)
        # tanh via v_exp/v_rcp (abs err ~2e-7) and MFMA summation order: states to 2e-5 absolute
        np.testing.assert_allclose(traj, to, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, np.array([0.3], np.float32)), rtol=5e-5, atol=1e-3)
        e.close()


@pytest.mark.parametrize("materialize", [True, False])   # False = the instantiation bench.py times
def test_mppi_mlp_matches_reference_golden(materialize):
    d = load("mppi_mlp.npz")
    e = mppi_engine_from(d, materialize=materialize)
    H = int(d["mpc_horizon"])
    for t in range(int(d["steps"])):
        u = e.step(d[f"s_{t}"], d[f"noise_{t}"], u_prev=[d[f"u_prev_{t}"]])
        if materialize:
            close(f"mppi_mlp[materialize={materialize}] step {t}", "q", e.read("Q"), d[f"u_run_{t}"], rtol=1e-6, atol=1e-6)
            close(f"mppi_mlp[materialize={materialize}] step {t}", "traj", e.read("TRAJ"), d[f"traj_{t}"], rtol=1e-4, atol=2e-5)
        close(f"mppi_mlp[materialize={materialize}] step {t}", "j", e.read("J"), d[f"J_{t}"], rtol=J_RTOL, atol=1e-3)
        close(f"mppi_mlp[materialize={materialize}] step {t}", "u_nom", e.read("U_NOM"), d[f"u_nom_{t}"], **GOLDEN_U_TOL)
        close(f"mppi_mlp[materialize={materialize}] step {t}", "u", u, d[f"u_{t}"], **GOLDEN_U_TOL)
        e.set_state(np.concatenate([d[f"u_nom_{t}"].reshape(H), d[f"u_{t}"].reshape(1)]))
    e.close()


@pytest.mark.parametrize("N,H,p", [(2048, 100, 10), (1000, 35, 1), (16, 5, 2), (70, 12, 5)])
def test_mppi_mlp_matches_oracle(N, H, p):
    env = O.EnvParams(terminal_weight=0.25)
    w = O.mlp_default_weights(1)
    pred = O.Predictor("MLP", dt=0.02, env=env, weights=w)
    o = O.MPPI(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "MLP", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  materialize_trajectories=True)
    apply_env(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(N)
    s = np.array([0.1, -0.2, 2.5, 0.7], np.float32)
    for t in range(2):
        noise = rng.standard_normal((N, o.P, 1)).astype(np.float32)
        uo = o.step(s, noise)
        ug = e.step(s, noise)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=5e-5, atol=1e-3)
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug[0], uo, **U_TOL)
        s = pred.step(s.reshape(1, 4), np.array([uo], np.float32))[0]
    e.close()


def test_cem_and_random_mlp_match_oracle():
    env = O.EnvParams()
    w = O.mlp_default_weights(2)
    pred = O.Predictor("MLP", dt=0.02, env=env, weights=w)
    N, H, K = 500, 20, 50
    o = O.CEM(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, cem_outer_it=2, cem_best_k=K)
    e = CtkEngine("cem", "MLP", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=2, cem_best_k=K)
    apply_env(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(3)
    s = np.array([0.0, 0.3, -1.0, 0.2], np.float32)
    for t in range(2):
        noise = rng.standard_normal((2, N, H, 1)).astype(np.float32)
        uo, ug = o.step(s, noise), e.step(s, noise)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=5e-5, atol=1e-3)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-5, atol=2e-6)
    e.close()
    r = O.RandomAction(pred, O.Cost(env), num_rollouts=100, mpc_horizon=10)
    g = CtkEngine("random_action", "MLP", num_rollouts=100, mpc_horizon=10, dt=0.02)
    apply_env(g, env); g.set_predictor_weights(w)
    u01 = rng.random((100, 10, 1), dtype=np.float32)
    np.testing.assert_array_equal(g.step(s, u01)[0], r.step(s, u01))
    np.testing.assert_allclose(g.read("J"), r.J, rtol=5e-5, atol=1e-3)
    g.close()


def test_mppi_cfg5_full_size_eight_shards_equal_one_handle():
    """BASELINE config 5 at full size (MPPI, N = 65536, H = 100, MLP): size-independent properties —
    8 shards of 8192 merged through the record exchange == one handle of 65536 (which takes the
    multi-launch hierarchical merge, > 256 blocks), inputs within limits, deterministic."""
    import torch
    N, H, p, G = 65536, 100, 10, 8
    w = O.mlp_default_weights(0)
    kw = dict(mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=5)
    full = CtkEngine("mppi", "MLP", num_rollouts=N, **kw)
    full.set_predictor_weights(w)
    sh = []
    for i in range(G):
        e = CtkEngine("mppi", "MLP", num_rollouts=N // G, global_rollout_offset=i * (N // G), **kw)
        e.set_predictor_weights(w)
        sh.append(e)
    rec = full.mppi_partial_size()
    parts = torch.zeros(G * rec, dtype=torch.float32, device="cuda")
    s = np.array([0.05, 0.0, 2.0, 0.5], np.float32)
    for t in range(2):
        u_full = full.step(s, None)                       # device Philox, addressed by global rollout index
        for i, e in enumerate(sh):
            e.mppi_step_begin(s, parts.data_ptr() + 4 * i * rec, None)
        torch.cuda.synchronize()
        us = [e.mppi_step_end(parts.data_ptr(), G) for e in sh]
        assert all(np.array_equal(us[0], u) for u in us[1:])
        np.testing.assert_allclose(us[0], u_full, **U_TOL)
        np.testing.assert_allclose(sh[3].read("U_NOM"), full.read("U_NOM"), **U_TOL)
        assert np.all(np.abs(full.read("U_NOM")) <= 1.0)
        J = full.read("J")
        assert np.isfinite(J).all() and J.shape == (N,)
        Js = np.concatenate([e.read("J") for e in sh])
        if t == 0:
            # same draws, same nominal plan; the 8192-rollout shards run the pair form of the network step (two waves per
            # tile, ctk_mlp.h: mlp_step_pair), the 65536-rollout handle the one-wave form: other association of the sums
            np.testing.assert_allclose(Js, J, rtol=3e-6)
        else:
            np.testing.assert_allclose(Js, J, rtol=1e-5)  # u_nom now differs by the merge's summation order
    for e in sh + [full]:
        e.close()


def test_single_wave_forms_also_match_oracle():
    """The sizes the oracle can check run the pair form (MPPI, N <= 8192) and the wide form (RPGD, N <= 4096) of the MLP
    kernels; the one-wave-per-tile MPPI kernel and the single-launch RPGD descent serve larger N.  Their diagnostic
    switches (read once per process) put them under the same oracle / golden tests in a child process."""
    import os, subprocess, sys
    env = dict(os.environ, CTK_MPPI_NO_PAIR="1", CTK_RPGD_NARROW="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_mlp.py"), os.path.join(here, "test_gpu_rpgd.py"),
                        "-k", "(mlp and (oracle or golden)) and not single_wave"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("materialize", [False, True])
def test_mppi_mlp_angle_beyond_the_fast_cos_range_takes_the_checked_pass(materialize):
    """A state angle beyond the unchecked cos's range (|x| > 32768 rad) in ONE tile of a workgroup: every wave of the workgroup
    must take the checked second pass together (workgroup barriers inside the pair step) — and the costs still match the oracle."""
    env = O.EnvParams(terminal_weight=0.25)
    w = O.mlp_default_weights(1)
    pred = O.Predictor("MLP", dt=0.02, env=env, weights=w)
    N, H, p = 96, 9, 3
    o = O.MPPI(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    e = CtkEngine("mppi", "MLP", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  materialize_trajectories=materialize)
    apply_env(e, env); e.set_predictor_weights(w)
    rng = np.random.default_rng(3)
    s = np.array([0.1, -0.2, 40000.25, 0.7], np.float32)         # the initial angle itself is out of range: every trajectory sees it
    noise = rng.standard_normal((N, o.P, 1)).astype(np.float32)
    uo, ug = o.step(s, noise), e.step(s, noise)
    np.testing.assert_allclose(e.read("J"), o.J, rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(ug[0], uo, rtol=2e-3, atol=2e-4)
    e.close()
