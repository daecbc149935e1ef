"""-m gpu: MPPI through the C ABI (libctk_hip.so) against the golden fixtures recorded from the
reference and against the oracle on seeded inputs."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from helpers import load, mppi_oracle_from, MPPI_CASES
from gpu_helpers import mppi_engine_from, apply_env
from margins import close

pytestmark = pytest.mark.gpu

# fp32 tolerances (stated here once):
#  * J: rtol 3e-5 — H+1 = 51 fp32 additions of terms up to ~2e4, different summation order/FMA use.
#  * u_nom / u: the soft-min turns a cost error dJ into a relative weight error dJ/LBD.  With
#    J ~ 1e4 (ep_weight 2e4) fp32 rounding gives dJ ~ 1e-2..1e-1, LBD = 100 => 1e-4..1e-3 relative
#    weight error on perturbations of size stdev = 0.21, averaged over N => a few 1e-6..1e-5 absolute.
U_TOL = dict(rtol=1e-4, atol=2e-5)
# Against the REFERENCE-RECORDED fixtures the observed errors are written to profiles/r04_parity_margins.txt (tests/margins.py) and the
# bounds are set from them (round 4): J worst 1.3e-6 relative -> rtol 1e-5 (SURVEY 8c's proposal); u_nom / u worst 4.7e-6 absolute
# (mppi_tiny_ode: N = 8, a handful of weights decides) and 9.1e-7 at cfg2 -> rtol 1e-5 / atol 1e-5.  SURVEY 8c's atol 1e-6 on u_nom
# does not hold at N = 8 for the reason in the comment above; U_TOL stays for the oracle-seeded cases whose margins are not recorded.
J_RTOL = 1e-5
GOLDEN_U_TOL = dict(rtol=1e-5, atol=1e-5)

ODE_CASES = [c for c in MPPI_CASES if c.endswith("_ode")]


# materialize=False is the instantiation bench.py times (ctk_mppi_rollout<PRED, false>: no u_run / trajectory stores);
# materialize=True the logging one.  Both meet the reference-recorded fixture, cfg2 at its full size (N 1024, H 50).
@pytest.mark.parametrize("materialize", [True, False])
@pytest.mark.parametrize("case", ODE_CASES)
def test_mppi_matches_reference_golden(case, materialize):
    d = load(f"mppi_{case}.npz")
    e = mppi_engine_from(d, materialize=materialize)
    assert e.dominant_kernel() == f"ctk_mppi_rollout<0, 0, {'true' if materialize else 'false'}, false>"
    H = int(d["mpc_horizon"])
    np.testing.assert_array_equal(e.read("U_NOM"), d["u_nom_init"])
    for t in range(int(d["steps"])):
        u = e.step(d[f"s_{t}"], d[f"noise_{t}"], u_prev=[d[f"u_prev_{t}"]])
        tag = f"mppi_{case}[materialize={materialize}] step {t}"
        if materialize:
            close(tag, "u_run", e.read("Q"), d[f"u_run_{t}"], rtol=1e-6, atol=1e-6)
        close(tag, "J", e.read("J"), d[f"J_{t}"], rtol=J_RTOL)
        close(tag, "u_nom", e.read("U_NOM"), d[f"u_nom_{t}"], **GOLDEN_U_TOL)
        close(tag, "u", u, d[f"u_{t}"], **GOLDEN_U_TOL)
        if materialize and f"traj_{t}" in d.files:
            close(tag, "traj", e.read("TRAJ"), d[f"traj_{t}"], rtol=1e-4, atol=2e-5)
        # re-pin to the reference's own warm-start state so every step is checked in isolation
        e.set_state(np.concatenate([d[f"u_nom_{t}"].reshape(H), d[f"u_{t}"].reshape(1)]))
    e.close()


def test_mppi_closed_loop_own_state_matches_golden():
    # no re-pinning: the engine's own u_nom / u carry over exactly like the reference's
    d = load("mppi_interp_ode.npz")
    e = mppi_engine_from(d, materialize=False)
    for t in range(int(d["steps"])):
        u = e.step(d[f"s_{t}"], d[f"noise_{t}"])        # u_prev = engine's own last output
        np.testing.assert_allclose(u, d[f"u_{t}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(e.read("U_NOM"), d[f"u_nom_{t}"], rtol=1e-4, atol=1e-5)
    e.close()


@pytest.mark.parametrize("N,H,p", [(1024, 50, 1), (1024, 50, 10), (1000, 35, 10), (70, 7, 3), (1, 1, 1), (130, 100, 10)])
def test_mppi_matches_oracle_seeded(N, H, p):
    d = load("mppi_tiny_ode.npz")
    pred = O.Predictor("ODE", dt=0.02, env=O.EnvParams(terminal_weight=0.3, target_position=0.05))
    o = O.MPPI(pred, O.Cost(pred.env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    from control_toolkit_amd import CtkEngine
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                  materialize_trajectories=True)
    apply_env(e, pred.env)
    rng = np.random.default_rng(N + H)
    s = np.array([0.1, -0.2, 2.5, 0.7], np.float32)
    for t in range(3):
        noise = rng.standard_normal((N, o.P, 1)).astype(np.float32)
        uo = o.step(s, noise)
        ug = e.step(s, noise)
        np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
        # states: rtol 1e-4 after H = 50 steps.  The upright pendulum is an unstable
        # equilibrium (e-folding time sqrt(l/g) ~ 0.14 s = 7 steps), so 1-ulp differences (FMA
        # contraction, reciprocal-based division) grow along the horizon: atol scales with H.
        np.testing.assert_allclose(e.read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=2e-5 * max(1, H // 25))
        np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
        np.testing.assert_allclose(ug[0], uo, **U_TOL)
        s = pred.step(s.reshape(1, 4), np.array([uo], np.float32))[0]
    e.close()


def test_mppi_properties_full_size():
    """BASELINE cfg2 size with size-independent properties: bounds, shift invariance of the
    update in J (via a cost offset none exists -> use duplicate-shard merge), determinism."""
    from control_toolkit_amd import CtkEngine
    N, H = 1024, 50
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=1)
    rng = np.random.default_rng(5)
    s = np.array([0.0, 0.0, 3.0, 0.0], np.float32)
    noise = rng.standard_normal((N, H, 1)).astype(np.float32)
    u1 = e.step(s, noise)
    un1 = e.read("U_NOM")
    assert np.all(np.abs(un1) <= 1.0) and np.all(np.abs(e.read("Q")) <= 1.0)     # optimizer_mppi.py:187,190
    e.reset(); e.set_state(np.zeros(H + 1, np.float32))
    u2 = e.step(s, noise)
    np.testing.assert_array_equal(u1, u2)                                            # deterministic
    np.testing.assert_array_equal(un1, e.read("U_NOM"))
    # weights sum to one: with all perturbations equal the update equals that perturbation
    e.reset(); e.set_state(np.zeros(H + 1, np.float32))
    const = np.full((N, H, 1), 0.25, np.float32)
    e.step(s, const)
    stdev = np.float32(0.03 / np.sqrt(0.02))
    np.testing.assert_allclose(e.read("U_NOM")[0, :, 0], np.full(H, 0.25 * stdev, np.float32), rtol=1e-5)
    e.close()


def test_mppi_device_rng_matches_oracle_philox():
    from control_toolkit_amd import CtkEngine
    N, H, p = 256, 20, 1
    e = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=1234,
                  materialize_trajectories=True, global_rollout_offset=512)
    pred = O.Predictor("ODE")
    o = O.MPPI(pred, O.Cost(pred.env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    s = np.array([0.0, 0.1, 1.0, -0.3], np.float32)
    for call in range(2):
        noise = O.device_noise(seed=1234, stream=0, call=call, first_row=512, rows=N, cols=H, kind="normal")
        uo = o.step(s, noise.reshape(N, H, 1))
        ug = e.step(s, None)
        np.testing.assert_allclose(e.read("Q"), o.u_run, rtol=1e-4, atol=2e-6)   # Box-Muller transcendental rounding
        np.testing.assert_allclose(ug[0], uo, rtol=1e-4, atol=1e-5)
    e.close()


def test_mppi_sharded_begin_end_equals_single_step():
    """two shards of 512 + merge of their records == one handle of 1024."""
    import torch
    from control_toolkit_amd import CtkEngine
    N, H, p = 1024, 50, 10
    kw = dict(mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p)
    full = CtkEngine("mppi", "ODE", num_rollouts=N, **kw)
    sh = [CtkEngine("mppi", "ODE", num_rollouts=N // 2, global_rollout_offset=i * N // 2, **kw) for i in range(2)]
    rng = np.random.default_rng(9)
    s = np.array([0.05, 0.0, 2.0, 0.5], np.float32)
    rec = full.mppi_partial_size()
    parts = torch.zeros(2 * rec, dtype=torch.float32, device="cuda")
    for t in range(3):
        P = rec - 2
        noise = rng.standard_normal((N, P, 1)).astype(np.float32)
        u_full = full.step(s, noise)
        for i, e in enumerate(sh):
            e.mppi_step_begin(s, parts.data_ptr() + 4 * i * rec, noise[i * N // 2:(i + 1) * N // 2])
        torch.cuda.synchronize()
        us = [e.mppi_step_end(parts.data_ptr(), 2) for e in sh]
        np.testing.assert_array_equal(us[0], us[1])
        np.testing.assert_allclose(us[0], u_full, **U_TOL)
        np.testing.assert_allclose(sh[0].read("U_NOM"), full.read("U_NOM"), **U_TOL)
    for e in sh + [full]:
        e.close()


def test_errors_are_loud():
    from control_toolkit_amd import CtkEngine
    with pytest.raises(ValueError):
        CtkEngine("mppi", "ODE", num_rollouts=0, mpc_horizon=10, dt=0.02)
    with pytest.raises(ValueError, match="num_states == 4"):     # S / C are the environment's (Optimizers/__init__.py:52-63 take the predictor's)
        CtkEngine("mppi", "ODE", num_rollouts=8, mpc_horizon=10, dt=0.02, num_states=6)
    with pytest.raises(NotImplementedError, match="not built"):
        CtkEngine("mppi", "ODE", num_rollouts=8, mpc_horizon=10, dt=0.02, environment="Acrobot")
    e = CtkEngine("mppi", "ODE", num_rollouts=8, mpc_horizon=10, dt=0.02)
    with pytest.raises(ValueError):
        e.step(np.zeros(4, np.float32), np.zeros((3, 10, 1), np.float32))
    with pytest.raises(Exception):
        e.read("TRAJ")      # not materialised
    e.close()


@pytest.mark.parametrize("N,H,p,isteps", [(40000, 12, 1, 1), (70001, 10, 3, 1), (131072, 6, 1, 1), (66000, 8, 2, 2)])
def test_mppi_throughput_variants_match_oracle(N, H, p, isteps):
    from control_toolkit_amd import CtkEngine
    """N >= 32768 takes the single-wave throughput kernel (inputs formed inline, half the LDS) and a tree of merge
    launches; against the oracle, logging on and off, ragged N, Euler sub-steps."""
    env = O.EnvParams(terminal_weight=0.2)
    pred = O.Predictor("ODE", dt=0.02, intermediate_steps=isteps, env=env)
    o = O.MPPI(pred, O.Cost(env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    engines = [CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                         intermediate_steps=isteps, materialize_trajectories=log) for log in (True, False)]
    for e in engines:
        apply_env(e, env)
    name = engines[1].dominant_kernel()
    assert "ctk_mppi_rollout_tp" in name
    rng = np.random.default_rng(N)
    s = np.array([0.1, -0.2, 2.5, 0.7], np.float32)
    for t in range(2):
        noise = rng.standard_normal((N, o.P, 1)).astype(np.float32)
        uo = o.step(s, noise)
        for e in engines:
            ug = e.step(s, noise)
            np.testing.assert_allclose(e.read("J"), o.J, rtol=3e-5)
            np.testing.assert_allclose(e.read("U_NOM"), o.u_nom, **U_TOL)
            np.testing.assert_allclose(ug[0], uo, **U_TOL)
        np.testing.assert_allclose(engines[0].read("TRAJ"), o.rollout_trajectories, rtol=1e-4, atol=1e-5 * H)
        np.testing.assert_allclose(engines[0].read("Q"), o.u_run, rtol=1e-6, atol=5e-6)   # interpolation: FMA vs the oracle's matmul
        s = pred.step(s.reshape(1, 4), np.array([uo], np.float32))[0]
    for e in engines:
        e.close()
