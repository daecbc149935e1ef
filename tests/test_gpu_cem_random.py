"""-m gpu: CEM and random-action through the C ABI against the oracle (their reference modules
import tensorflow at module level and cannot be executed here: the oracle follows the source
text of optimizer_cem_tf.py / optimizer_random_action_tf.py — parity unpinned by a reference run)."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from gpu_helpers import apply_env

pytestmark = pytest.mark.gpu


def make_cem(N, H, K, its, env, mat=True, **kw):
    pred = O.Predictor("ODE", dt=0.02, env=env)
    o = O.CEM(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, cem_outer_it=its, cem_best_k=K, **kw)
    e = CtkEngine("cem", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=its, cem_best_k=K,
                  cem_initial_action_stdev=kw.get("cem_initial_action_stdev", 0.5), cem_stdev_min=kw.get("cem_stdev_min", 0.01),
                  warmup=int(kw.get("warmup", False)), warmup_iterations=kw.get("warmup_iterations", 250),
                  materialize_trajectories=mat)
    apply_env(e, env)
    return pred, o, e


# every size up to 128 workgroups runs the ONE-launch step (ctk_cem_fused.hip; <true> with trajectories, <false> without — the
# benchmarked instantiation); (16384, ...) is beyond it and takes the launch-per-phase form
@pytest.mark.parametrize("mat", [True, False])
@pytest.mark.parametrize("N,H,K,its", [(4096, 30, 409, 3), (200, 40, 40, 3), (64, 8, 64, 1), (100, 5, 1, 2), (16384, 12, 900, 2)])
def test_cem_matches_oracle(N, H, K, its, mat):
    env = O.EnvParams(terminal_weight=0.2)
    pred, o, e = make_cem(N, H, K, its, env, mat=mat)
    assert e.dominant_kernel().startswith("ctk_cem_fused") == (N <= 8192)
    rng = np.random.default_rng(N)
    s = np.array([0.02, 0.1, 2.9, -0.5], np.float32)
    for t in range(3):
        noise = rng.standard_normal((its, N, H, 1)).astype(np.float32)
        uo = o.step(s, noise)
        ug = e.step(s, noise)
        Jg = e.read("J")
        np.testing.assert_allclose(Jg, o.J, rtol=3e-5)
        np.testing.assert_allclose(e.read("Q"), o.Q, rtol=1e-5, atol=2e-6)
        # elite set: identical unless two costs are closer than the fp32 cost tolerance at the cut
        bg = e.read("BEST_IDX")
        srt = np.sort(o.J)
        gap_ok = (srt[K] - srt[K - 1]) > 1e-4 * abs(srt[K - 1]) if K < N else True
        if gap_ok:
            assert set(bg.tolist()) == set(o.best_idx.tolist())
        assert np.all(np.diff(Jg[bg]) >= 0)                      # sorted ascending (tf.argsort)
        assert bg[0] == np.argmin(Jg)
        np.testing.assert_allclose(e.read("U_NOM"), o.dist_mue, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(e.read("STD"), o.stdev, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ug[0], uo, rtol=1e-5, atol=2e-6)
        if mat:
            np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=4e-5)
        s = pred.step(s.reshape(1, 4), np.array([uo], np.float32))[0]
    e.close()


def _cem_pair(monkeypatch, **kw):
    """the same configuration as the one-launch step and (CTK_NO_CEM_FUSED at creation) as the launch-per-phase form"""
    fused = CtkEngine("cem", "ODE", **kw)
    monkeypatch.setenv("CTK_NO_CEM_FUSED", "1")
    plain = CtkEngine("cem", "ODE", **kw)
    monkeypatch.delenv("CTK_NO_CEM_FUSED")
    assert fused.dominant_kernel().startswith("ctk_cem_fused") and plain.dominant_kernel().startswith("ctk_affine_rollout")
    return fused, plain


@pytest.mark.parametrize("N,H,K,its", [(4096, 30, 409, 3), (1000, 17, 77, 4), (8192, 10, 8000, 2)])
def test_cem_one_launch_equals_launch_per_phase_device_rng(monkeypatch, N, H, K, its):
    """in-kernel sampler (Philox keyed by global row / iteration): both forms see the same draws; closed loop over 5 steps"""
    fused, plain = _cem_pair(monkeypatch, num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=its, cem_best_k=K, seed=7)
    pred = O.Predictor("ODE", dt=0.02, env=O.EnvParams())
    s = np.array([0.02, 0.1, 2.9, -0.5], np.float32)
    for t in range(5):
        uf, up = fused.step(s), plain.step(s)
        np.testing.assert_allclose(uf, up, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(fused.read("U_NOM"), plain.read("U_NOM"), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(fused.read("STD"), plain.read("STD"), rtol=5e-5, atol=1e-6)
        np.testing.assert_allclose(fused.read("J"), plain.read("J"), rtol=3e-5)
        np.testing.assert_array_equal(fused.read("BEST_IDX")[0], plain.read("BEST_IDX")[0])
        s = pred.step(s.reshape(1, 4), np.asarray(up, np.float32))[0]
    fused.close(); plain.close()


@pytest.mark.parametrize("K", [31, 32, 1, 255])
def test_cem_one_launch_breaks_cost_ties_by_index(monkeypatch, K):
    """every plan appears twice (rows n and n + 128): each cost is an exact tie and an odd K cuts a pair — the elite set must be
    the first K of (cost, index), as tf.argsort / ctk_select_topk give it; a wrong tie rule changes the refit visibly"""
    N, H = 256, 9
    fused, plain = _cem_pair(monkeypatch, num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=2, cem_best_k=K)
    rng = np.random.default_rng(K)
    noise = rng.standard_normal((2, N, H, 1)).astype(np.float32)
    noise[:, N // 2:] = noise[:, : N // 2]
    s = np.array([0.0, 0.0, 1.0, 0.0], np.float32)
    uf, up = fused.step(s, noise), plain.step(s, noise)
    J = fused.read("J")
    assert np.array_equal(J[: N // 2], J[N // 2:])
    np.testing.assert_array_equal(fused.read("BEST_IDX"), np.argsort(J, kind="stable")[:K])
    np.testing.assert_allclose(fused.read("U_NOM"), plain.read("U_NOM"), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(fused.read("STD"), plain.read("STD"), rtol=5e-5, atol=1e-6)
    np.testing.assert_allclose(uf, up, rtol=1e-6, atol=1e-7)
    fused.close(); plain.close()


@pytest.mark.parametrize("N,K,groups", [(2048, 100, 1), (2048, 700, 1), (4096, 409, 3), (1536, 1000, 2)])
def test_cem_one_launch_with_many_equal_costs(monkeypatch, N, K, groups):
    """`groups` distinct plans, each repeated N / groups times: more equal costs than the selection's second stage holds, so
    the one-launch step takes its radix-pass fallback and ranks the ties by index; elite set == first K of (cost, index)"""
    H = 6
    fused, plain = _cem_pair(monkeypatch, num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=2, cem_best_k=K)
    rng = np.random.default_rng(N + K)
    base = rng.standard_normal((2, groups, H, 1)).astype(np.float32)
    noise = np.ascontiguousarray(base[:, np.arange(N) % groups])
    s = np.array([0.0, 0.0, 1.0, 0.0], np.float32)
    uf, up = fused.step(s, noise), plain.step(s, noise)
    J = fused.read("J")
    assert len(np.unique(J)) <= groups
    np.testing.assert_array_equal(fused.read("BEST_IDX"), np.argsort(J, kind="stable")[:K])
    np.testing.assert_allclose(fused.read("U_NOM"), plain.read("U_NOM"), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(fused.read("STD"), plain.read("STD"), rtol=5e-5, atol=1e-6)
    np.testing.assert_allclose(uf, up, rtol=1e-6, atol=1e-7)
    fused.close(); plain.close()


def test_cem_warmup_and_reset():
    env = O.EnvParams()
    pred, o, e = make_cem(128, 10, 16, 2, env, warmup=True, warmup_iterations=5)
    rng = np.random.default_rng(1)
    s = np.array([0.0, 0.0, 3.0, 0.0], np.float32)
    assert e.samples_needed() == 5 * 128 * 10            # first step runs warmup_iterations (optimizer_cem_tf.py:92)
    n0 = rng.standard_normal((5, 128, 10, 1)).astype(np.float32)
    np.testing.assert_allclose(e.step(s, n0)[0], o.step(s, n0), rtol=1e-5, atol=2e-6)
    assert e.samples_needed() == 2 * 128 * 10
    n1 = rng.standard_normal((2, 128, 10, 1)).astype(np.float32)
    np.testing.assert_allclose(e.step(s, n1)[0], o.step(s, n1), rtol=1e-5, atol=2e-6)
    e.reset(); o.optimizer_reset()
    np.testing.assert_array_equal(e.read("U_NOM"), o.dist_mue)
    np.testing.assert_array_equal(e.read("STD"), o.stdev)
    assert e.samples_needed() == 5 * 128 * 10
    e.close()


def test_cem_best_k_equals_n_gives_population_mean():
    env = O.EnvParams()
    N, H = 256, 12
    _, o, e = make_cem(N, H, N, 1, env)
    noise = np.random.default_rng(2).standard_normal((1, N, H, 1)).astype(np.float32)
    s = np.array([0.1, 0.0, 0.5, 0.0], np.float32)
    e.step(s, noise)
    Q = e.read("Q")
    # mu (before the shift) == mean(Q)  (SURVEY 4); after the shift mu[h] == mean(Q)[h+1]
    np.testing.assert_allclose(e.read("U_NOM")[0, :-1, 0], Q.mean(0)[1:, 0], rtol=1e-5, atol=1e-6)
    assert e.read("U_NOM")[0, -1, 0] == 0.0 and e.read("STD")[0, -1, 0] == np.float32(0.5)
    e.close()


@pytest.mark.parametrize("N,H", [(32, 10), (320, 35), (1000, 20)])
def test_random_action_matches_oracle(N, H):
    # (32, 10) is BASELINE config 1
    env = O.EnvParams(terminal_weight=0.1)
    pred = O.Predictor("ODE", dt=0.02, env=env)
    o = O.RandomAction(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H)
    e = CtkEngine("random_action", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, materialize_trajectories=True)
    apply_env(e, env)
    rng = np.random.default_rng(N)
    s = np.array([0.0, 0.1, 0.5, -0.2], np.float32)
    for t in range(3):
        u01 = rng.random((N, H, 1), dtype=np.float32)
        uo = o.step(s, u01)
        ug = e.step(s, u01)
        np.testing.assert_allclose(e.read("Q"), o.Q, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        assert int(e.read("BEST_IDX")[0]) == int(o.best_idx)
        np.testing.assert_array_equal(ug[0], uo)
        s = pred.step(s.reshape(1, 4), np.array([uo], np.float32))[0]
    e.close()


def test_device_rng_cem_and_random_run():
    e = CtkEngine("cem", "ODE", num_rollouts=512, mpc_horizon=20, dt=0.02, cem_outer_it=2, cem_best_k=50, seed=3)
    s = np.array([0.0, 0.0, 3.0, 0.0], np.float32)
    u = e.step(s)
    Q = e.read("Q")
    assert np.isfinite(u).all() and abs(Q.mean()) < 0.1 and 0.2 < Q.std() < 0.7
    noise_like = O.device_noise(seed=3, stream=1, call=0, first_row=0, rows=512, cols=20, kind="normal")
    assert noise_like.shape == (512, 20)
    e.close()
    r = CtkEngine("random_action", "ODE", num_rollouts=256, mpc_horizon=10, dt=0.02, seed=4)
    u = r.step(s)
    Q = r.read("Q")
    exp = O.device_noise(seed=4, stream=0, call=0, first_row=0, rows=256, cols=10, kind="uniform") * 2 - 1
    np.testing.assert_allclose(Q[:, :, 0], exp, rtol=0, atol=1e-6)
    assert u[0] == Q[np.argmin(r.read("J")), 0, 0]
    r.close()


def test_plain_rollout_matches_oracle():
    env = O.EnvParams(terminal_weight=0.4)
    pred = O.Predictor("ODE", dt=0.02, env=env)
    cost = O.Cost(env)
    e = CtkEngine("mppi", "ODE", num_rollouts=64, mpc_horizon=25, dt=0.02)
    apply_env(e, env)
    Q = np.random.default_rng(0).uniform(-1.5, 1.5, (10, 25, 1)).astype(np.float32)   # beyond the limits: taken as given
    s = np.array([0.1, 0.2, 1.0, -1.0], np.float32)
    traj, J = e.rollout(s, Q, u_prev=0.3)
    to = pred.predict_core(np.tile(s, (10, 1)), Q)
    np.testing.assert_allclose(traj, to, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(J, cost.get_trajectory_cost(to, Q, np.array([0.3], np.float32)), rtol=3e-5)
    e.close()


@pytest.mark.parametrize("opt", ["cem", "random_action"])
def test_sharded_topk_two_shards_equal_one_handle(opt):
    """two shards of N/2 + one exchange of best-K records per iteration == one handle of N."""
    import torch
    N, H, K, its = 2048, 30, 205, 3
    kw = dict(cem_outer_it=its, cem_best_k=K) if opt == "cem" else {}
    full = CtkEngine(opt, "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, **kw)
    sh = [CtkEngine(opt, "ODE", num_rollouts=N // 2, mpc_horizon=H, dt=0.02, global_rollout_offset=i * N // 2, **kw) for i in range(2)]
    rec = sh[0].shard_candidates_size()
    assert rec == (K if opt == "cem" else 1) * (2 + H)
    gathered = torch.zeros(2 * rec, dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(4)
    s = np.array([0.02, 0.1, 2.9, -0.5], np.float32)
    n_it = sh[0].shard_iterations()
    assert n_it == (its if opt == "cem" else 1)
    for t in range(3):
        draws = (rng.standard_normal((n_it, N, H, 1)) if opt == "cem" else rng.random((n_it, N, H, 1))).astype(np.float32)
        u_full = full.step(s, draws if opt == "cem" else draws[0])
        for it in range(n_it):
            for i, e in enumerate(sh):
                e.shard_iter_begin(s, gathered.data_ptr() + 4 * i * rec, draws[it, i * N // 2:(i + 1) * N // 2])
            torch.cuda.synchronize()
            for e in sh:
                e.shard_iter_end(gathered.data_ptr(), 2)
        us = [e.shard_finish() for e in sh]
        np.testing.assert_array_equal(us[0], us[1])
        np.testing.assert_allclose(us[0], u_full, rtol=1e-6, atol=1e-7)
        if opt == "cem":
            np.testing.assert_allclose(sh[0].read("U_NOM"), full.read("U_NOM"), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(sh[1].read("STD"), full.read("STD"), rtol=1e-4, atol=1e-6)
    for e in sh + [full]:
        e.close()


@pytest.mark.parametrize("N,K", [(65536, 100), (5000, 5000), (1, 1), (63, 7)])
def test_select_topk_against_numpy(N, K):
    """the selection kernel alone, through random-action/CEM handles: sorted ascending, ties by index"""
    H = 2
    e = CtkEngine("cem", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, cem_outer_it=1, cem_best_k=K)
    rng = np.random.default_rng(N)
    noise = rng.standard_normal((1, N, H, 1)).astype(np.float32)
    noise[0, N // 2:, :, :] = noise[0, : N - N // 2, :, :][: N - N // 2]     # duplicate plans -> exact cost ties
    e.step(np.array([0.0, 0.0, 1.0, 0.0], np.float32), noise)
    J = e.read("J")
    got = e.read("BEST_IDX")
    np.testing.assert_array_equal(got, np.argsort(J, kind="stable")[:K])
    e.close()


@pytest.mark.parametrize("opt", ["mppi", "cem", "random_action", "rpgd"])
def test_degenerate_sizes_n1_h1(opt):
    kw = dict(cem=dict(cem_outer_it=2, cem_best_k=1), rpgd=dict(outer_its=2, opt_keep_k=1, resamp_per=1)).get(opt, {})
    e = CtkEngine(opt, "ODE", num_rollouts=1, mpc_horizon=1, dt=0.02, seed=2, **kw)
    if opt == "rpgd":
        e.reset()
    s = np.array([0.0, 0.0, 0.5, 0.0], np.float32)
    for _ in range(3):
        u = e.step(s)
        assert np.isfinite(u).all() and -1.0 <= u[0] <= 1.0
    e.close()
