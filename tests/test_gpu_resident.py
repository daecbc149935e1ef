"""-m gpu: the resident form of the MPPI step (include/ctk_hip.h: ctk_resident_*; csrc/ctk_mppi.hip: ctk_mppi_resident) — the same step
served from a pinned mailbox by a kernel that stays on the device.  It must be a pure latency optimisation: results bit-identical to the
launched form (the same statements, ctk_mppi_body.inc), every wait bounded, the device released when idle and at once on any other API call."""
import time

import numpy as np
import pytest

from oracle import ctk_oracle as O
from control_toolkit_amd import CtkEngine
from helpers import load, env_from
from test_gpu_mppi import U_TOL

pytestmark = pytest.mark.gpu
S0 = np.array([0.05, -0.1, 2.8, 0.4], np.float32)


def _pair(**kw):
    return CtkEngine("mppi", "ODE", **kw), CtkEngine("mppi", "ODE", **kw)


@pytest.mark.parametrize("N,H,p", [(1024, 50, 1), (300, 35, 10), (64, 7, 3), (8192, 20, 5)])
def test_resident_steps_equal_launched_steps_bit_for_bit(N, H, p):
    import torch
    a, b = _pair(num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=3)
    b.resident_enable(True, idle_us=100000.0)
    P = a.inducing_points()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    pool = [torch.randn((N * P,), generator=g, device="cuda") for _ in range(4)]
    s = S0.copy()
    for t in range(12):
        buf = pool[t & 3].data_ptr() if t % 3 else None           # device buffers and the in-kernel sampler, interleaved
        up = None if t % 2 else np.array([0.1 * np.sin(t)], np.float32)
        ua, ub = a.step(s, buf, u_prev=up), b.step(s, buf, u_prev=up)
        np.testing.assert_array_equal(ub, ua)
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    st = b.resident_stats()
    assert st["steps"] == 12 and st["launches"] == 1 and st["running"]
    np.testing.assert_array_equal(b.read("U_NOM"), a.read("U_NOM"))   # the read ends the resident kernel first
    np.testing.assert_array_equal(b.read("J"), a.read("J"))
    assert not b.resident_stats()["running"]
    ua, ub = a.step(s), b.step(s)                                     # ... and the next step launches it again
    np.testing.assert_array_equal(ub, ua)
    assert b.resident_stats()["launches"] == 2
    a.close(); b.close()


def test_resident_matches_reference_golden_at_the_benchmarked_size():
    d = load("mppi_cfg2_ode.npz")
    e = CtkEngine("mppi", "ODE", num_rollouts=int(d["num_rollouts"]), mpc_horizon=int(d["mpc_horizon"]), dt=float(d["dt"]),
                  period_interpolation_inducing_points=int(d["period_interpolation_inducing_points"]), cc_weight=float(d["cc_weight"]), R=float(d["R"]),
                  LBD=float(d["LBD"]), NU=float(d["NU"]), SQRTRHOINV=float(d["SQRTRHOINV"]))
    for n in env_from(d).param_names():
        e.set_param(n, float(getattr(env_from(d), n)))
    e.resident_enable(True, idle_us=50000.0)
    import torch
    H = int(d["mpc_horizon"])
    for t in range(int(d["steps"])):
        noise = torch.tensor(d[f"noise_{t}"], device="cuda")
        u = e.step(d[f"s_{t}"], noise.data_ptr(), u_prev=d[f"u_prev_{t}"])
        np.testing.assert_allclose(u, np.asarray(d[f"u_{t}"]).reshape(-1), **U_TOL)
        np.testing.assert_allclose(e.read("J"), d[f"J_{t}"], rtol=3e-5)
        np.testing.assert_allclose(e.read("U_NOM"), d[f"u_nom_{t}"], **U_TOL)
        e.set_state(np.concatenate([d[f"u_nom_{t}"].reshape(H), np.asarray(d[f"u_{t}"]).reshape(1)]))
    assert e.resident_stats()["steps"] == int(d["steps"])
    e.close()


def test_resident_refilled_buffer_is_read_at_the_request():
    """ONE device buffer refilled IN PLACE between steps (torch's `normal_()`), the way a caller without a pool would feed
    the sampler.  The default resident mode must read it when the request arrives — bit-identical to the launched form, step after step —
    and must not have prepared its inputs ahead from the buffer's previous contents (the t_relay flag bit says which path served)."""
    import torch
    N, H, p = 1024, 50, 1
    a, b = _pair(num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=3)
    b.resident_enable(True, idle_us=100000.0)                      # default: no read-ahead of caller-owned buffers
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    buf = torch.empty((N * a.inducing_points(),), device="cuda")
    s = S0.copy()
    for t in range(10):
        buf.normal_(generator=g)                                   # same pointer, new contents
        torch.cuda.current_stream().synchronize()                  # (torch's stream only: a device-wide synchronize would wait for the resident kernel's idle time-out)
        ua, ub = a.step(s, buf.data_ptr()), b.step(s, buf.data_ptr())
        np.testing.assert_array_equal(ub, ua)
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    assert b.resident_stats()["steps"] == 10 and b.resident_stats()["launches"] == 1
    np.testing.assert_array_equal(b.read("U_NOM"), a.read("U_NOM"))
    np.testing.assert_array_equal(b.read("J"), a.read("J"))
    a.close(); b.close()


def test_resident_read_ahead_is_an_opt_in_for_static_pools():
    """read_ahead=True is the caller's promise that the pool's contents do not change: same results, inputs prepared between steps"""
    import torch
    N, H = 1024, 50
    a, b = _pair(num_rollouts=N, mpc_horizon=H, dt=0.02, seed=3)
    b.resident_enable(True, idle_us=100000.0, read_ahead=True)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    pool = [torch.randn((N * a.inducing_points(),), generator=g, device="cuda") for _ in range(4)]
    s = S0.copy()
    for t in range(12):
        np.testing.assert_array_equal(b.step(s, pool[t & 3].data_ptr()), a.step(s, pool[t & 3].data_ptr()))
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    with pytest.raises(ValueError):
        b._check(b._lib.ctk_resident_enable(b._h, 3, 100.0))
    a.close(); b.close()


def test_resident_kernel_leaves_when_idle_and_comes_back():
    import torch
    a, b = _pair(num_rollouts=512, mpc_horizon=20, dt=0.02, period_interpolation_inducing_points=5, seed=5)
    b.resident_enable(True, idle_us=300.0)
    s = S0.copy()
    for t in range(5):
        np.testing.assert_array_equal(b.step(s), a.step(s))
    assert b.resident_stats()["running"]
    time.sleep(0.02)                                   # 20 ms without a request: the kernel has left by itself
    t0 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert dt < 5e-3, f"device-wide synchronize took {dt * 1e3:.2f} ms with an idle resident handle"
    for t in range(5):                                  # the next step notices, launches again, same results
        np.testing.assert_array_equal(b.step(s), a.step(s))
        time.sleep(0.002)                               # every gap is longer than the idle time: a launch per step, results unchanged
    st = b.resident_stats()
    assert st["steps"] == 10 and 2 <= st["launches"] <= 7
    a.close(); b.close()


def test_a_device_wide_synchronize_waits_at_most_the_idle_time():
    import torch
    e = CtkEngine("mppi", "ODE", num_rollouts=1024, mpc_horizon=50, dt=0.02, seed=6)
    e.resident_enable(True, idle_us=2000.0)
    e.step(S0); e.step(S0)
    t0 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0     # nobody told the kernel to stop
    assert dt < 0.05, f"{dt * 1e3:.1f} ms"
    e.step(S0); e.resident_stop()
    t0 = time.perf_counter(); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0    # after resident_stop: nothing left to wait for
    assert dt2 < 1e-3, f"{dt2 * 1e6:.0f} us"
    e.close()


def test_resident_enable_is_refused_where_it_does_not_apply_and_other_paths_keep_working():
    for kw in (dict(optimizer="cem"), dict(predictor="MLP"), dict(materialize_trajectories=True), dict(num_rollouts=65536)):
        opt, pred = kw.pop("optimizer", "mppi"), kw.pop("predictor", "ODE")
        args = dict(num_rollouts=256, mpc_horizon=10, dt=0.02); args.update(kw)
        e = CtkEngine(opt, pred, **args)
        with pytest.raises(NotImplementedError):
            e.resident_enable(True)
        e.close()
    e = CtkEngine("mppi", "ODE", num_rollouts=256, mpc_horizon=10, dt=0.02, seed=2)
    with pytest.raises(ValueError):
        e.resident_enable(True, idle_us=0.0)
    e.resident_enable(True, idle_us=1000.0)
    u1 = e.step(S0)
    noise = np.random.default_rng(0).standard_normal((256, e.inducing_points(), 1)).astype(np.float32)
    u2 = e.step(S0, noise)                              # HOST samples: served by the launched form (the resident kernel is ended first)
    assert np.isfinite(u1).all() and np.isfinite(u2).all() and not e.resident_stats()["running"]
    e.set_param("target_position", 0.1)                 # parameters are baked into the resident launch: a change takes effect on the next step
    u3 = e.step(S0)
    o = CtkEngine("mppi", "ODE", num_rollouts=256, mpc_horizon=10, dt=0.02, seed=2)
    o.step(S0); o.step(S0, noise); o.set_param("target_position", 0.1)
    np.testing.assert_array_equal(u3, o.step(S0))
    e.resident_enable(False)
    assert np.isfinite(e.step(S0)).all() and not e.resident_stats()["running"]
    e.close(); o.close()


def test_resident_on_the_other_environments():
    for envname, S in (("Quad2D", 6), ("Hover", 7)):
        kw = dict(environment=envname, num_rollouts=512, mpc_horizon=20, dt=0.02, period_interpolation_inducing_points=5, seed=8)
        a, b = CtkEngine("mppi", "ODE", **kw), CtkEngine("mppi", "ODE", **kw)
        b.resident_enable(True, idle_us=20000.0)
        s = np.linspace(0.1, 0.4, S).astype(np.float32)
        for t in range(6):
            np.testing.assert_array_equal(b.step(s), a.step(s))
            s[0] += 0.01
        assert b.resident_stats()["launches"] == 1
        a.close(); b.close()


def test_requests_racing_the_idle_exit_are_never_lost():
    """gaps between steps drawn around the idle time: the kernel's leave announcement and the host's next request cross in flight again and
    again — every step must still be served (by the leaving kernel's re-read, or by the relaunch), bit-identical to the launched twin"""
    rng = np.random.default_rng(0)
    a, b = _pair(num_rollouts=256, mpc_horizon=12, dt=0.02, period_interpolation_inducing_points=3, seed=11)
    b.resident_enable(True, idle_us=60.0)
    s = S0.copy()
    t_start = time.perf_counter()
    for t in range(1500):
        gap = rng.uniform(0.0, 140e-6)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < gap:
            pass
        ua, ub = a.step(s), b.step(s)
        assert np.array_equal(ua, ub), t
        s[0] = 0.05 + 0.01 * np.sin(0.1 * t)
    st = b.resident_stats()
    assert st["steps"] == 1500 and 1 < st["launches"] < 1500, st          # some gaps were short enough to stay, some long enough to leave
    assert time.perf_counter() - t_start < 20.0
    a.close(); b.close()
