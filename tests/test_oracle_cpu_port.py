"""CPU (no GPU): oracle/ctk_cpu.c — the native C + OpenMP restatement bench.py times as `cpu_baseline` — against the reference-recorded
MPPI goldens (tests/golden/mppi_*.npz, CartPole analytic predictor) and against the NumPy oracle for the random-action step.
Tolerances as for the HIP kernels (test_gpu_mppi.py): another fp32 evaluation with another summation order and another libm."""
import numpy as np
import pytest

from oracle import ctk_oracle as O
from oracle import ctk_cpu
from helpers import load, env_from

ODE_CASES = ["tiny_ode", "interp_ode", "cfg2_ode", "quirk_ode"]


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("case", ODE_CASES)
def test_c_port_mppi_matches_reference_golden(case, threads):
    d = load(f"mppi_{case}.npz")
    assert str(d["predictor"]) == "ODE"
    c = ctk_cpu.MppiCpu(env_from(d), float(d["dt"]), float(np.asarray(d["low"]).reshape(-1)[0]), float(np.asarray(d["high"]).reshape(-1)[0]), num_rollouts=int(d["num_rollouts"]),
                        mpc_horizon=int(d["mpc_horizon"]), cc_weight=float(d["cc_weight"]), R=float(d["R"]), LBD=float(d["LBD"]), NU=float(d["NU"]),
                        SQRTRHOINV=float(d["SQRTRHOINV"]), period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]),
                        threads=threads)
    np.testing.assert_array_equal(c.u_nom, d["u_nom_init"].reshape(-1))
    for t in range(int(d["steps"])):
        c.u = np.float32(np.asarray(d[f"u_prev_{t}"]).reshape(-1)[0])
        u = c.step(d[f"s_{t}"], d[f"noise_{t}"])
        np.testing.assert_allclose(c.J, d[f"J_{t}"], rtol=3e-5)
        np.testing.assert_allclose(c.u_nom, d[f"u_nom_{t}"].reshape(-1), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(u, np.asarray(d[f"u_{t}"]).reshape(-1), rtol=1e-4, atol=2e-5)
        c.u_nom = d[f"u_nom_{t}"].reshape(-1).astype(np.float32).copy()      # replay the recorded state, as the GPU golden tests do


def test_c_port_random_action_matches_oracle_cfg1():
    """BASELINE configs[0]: random-action N = 32, H = 10 — the reference's own CPU-runnable case"""
    env = O.EnvParams(terminal_weight=0.2)
    N, H = 32, 10
    o = O.RandomAction(O.Predictor("ODE", env=env), O.Cost(env), num_rollouts=N, mpc_horizon=H)
    rng = np.random.default_rng(5)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(3):
        u01 = rng.random((N, H, 1), dtype=np.float32)
        up = float(o.u)
        uo = o.step(s, u01)
        u, J, best = ctk_cpu.random_action_step(env, 0.02, -1.0, 1.0, s, up, u01[:, :, 0], threads=2)
        np.testing.assert_allclose(J, o.J, rtol=3e-5)
        assert best == int(o.best_idx) and u == np.float32(np.asarray(uo).reshape(-1)[0])
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)


def test_c_port_is_thread_count_invariant_to_rounding():
    env = O.EnvParams()
    N, H = 256, 30
    noise = np.random.default_rng(1).standard_normal((N, O.num_inducing_points(H, 5))).astype(np.float32)
    s = np.array([0.0, 0.1, 3.0, -0.2], np.float32)
    outs = []
    for th in (1, 3, 8):
        c = ctk_cpu.MppiCpu(env, 0.02, num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=5, threads=th)
        c.step(s, noise)
        outs.append((c.J.copy(), c.u_nom.copy()))
    for J, un in outs[1:]:
        np.testing.assert_array_equal(J, outs[0][0])                       # per-trajectory work does not depend on the thread count
        np.testing.assert_allclose(un, outs[0][1], rtol=1e-6, atol=1e-7)   # the reduction's association does (double accumulators)
