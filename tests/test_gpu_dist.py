"""-m gpu: the sharded MPPI path with the REAL engine and a real process group — two processes on the
one GPU of the test box, gloo backend on CUDA tensors (RCCL refuses two ranks on one device; the
collective call site is the same `all_gather_into_tensor`).  Result must equal one handle of 2x the size."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N_local, H, p, steps, seed, q, exchange="rccl"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from control_toolkit_amd import CtkEngine
    from control_toolkit_amd.dist import ShardedMPPI
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = CtkEngine("mppi", "ODE", num_rollouts=N_local, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                    seed=seed, global_rollout_offset=rank * N_local)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    sh = ShardedMPPI(eng, rank, world, device=torch.device("cuda", 0), exchange=exchange)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    us = []
    P = eng.mppi_partial_size() - 2
    rng = np.random.default_rng(seed)
    for t in range(steps):
        torch.cuda.synchronize()
        if exchange == "p2p":    # explicit draws: this rank's rows of the global noise
            noise = rng.standard_normal((world * N_local, P, 1)).astype(np.float32)[rank * N_local:(rank + 1) * N_local]
            us.append(float(sh.step(s, noise)[0]))
        else:
            us.append(float(sh.step(s, None)[0]))        # device Philox addressed by GLOBAL rollout index
    q.put((rank, us, eng.read("U_NOM"), sh.exchange, sh.p2p_error))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_process_sharded_mppi_equals_single_handle():
    from control_toolkit_amd import CtkEngine
    N, H, p, steps, seed = 2048, 40, 10, 3, 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N // 2, H, p, steps, seed, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    full = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=seed)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    ref = [float(full.step(s, None)[0]) for _ in range(steps)]
    np.testing.assert_array_equal(res[0][1], res[1][1])                       # identical on both ranks
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-4, atol=2e-5)          # == unsharded (same global draws)
    np.testing.assert_allclose(res[0][2], full.read("U_NOM"), rtol=1e-4, atol=2e-5)
    full.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("N,H,p", [(2048, 40, 10), (16384, 20, 1)])
def test_two_process_p2p_exchange_equals_single_handle(N, H, p):
    """The peer-to-peer exchange (ctk_p2p_*: IPC-mapped uncached buffers, records stored straight into the peer's
    memory, flags, bounded wait) between two processes — here both on the one GPU of the test box; the set-up
    self-test must have accepted it (no silent fallback), and the result must equal one handle of 2x the size.
    N = 16384 takes the multi-launch reduction before the exchange (128 block records per shard)."""
    from control_toolkit_amd import CtkEngine
    steps, seed = 5, 78
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + ((os.getpid() + 7 + N) % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N // 2, H, p, steps, seed, q, "p2p")) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert res[0][3] == "p2p" and res[1][3] == "p2p", f"fell back to rccl: {res[0][4]} / {res[1][4]}"
    full = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=seed)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    P = full.mppi_partial_size() - 2
    rng = np.random.default_rng(seed)
    ref = [float(full.step(s, rng.standard_normal((N, P, 1)).astype(np.float32))[0]) for _ in range(steps)]
    np.testing.assert_array_equal(res[0][1], res[1][1])                       # identical on both ranks
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(res[0][2], full.read("U_NOM"), rtol=1e-4, atol=2e-5)
    full.close()


def test_p2p_single_rank_and_error_paths():
    """world = 1 degenerates to a local merge; API misuse is loud."""
    from control_toolkit_amd import CtkEngine
    e = CtkEngine("mppi", "ODE", num_rollouts=512, mpc_horizon=20, dt=0.02, seed=3)
    f = CtkEngine("mppi", "ODE", num_rollouts=512, mpc_horizon=20, dt=0.02, seed=3)
    s = np.array([0.0, 0.1, 0.5, 0.0], np.float32)
    with pytest.raises(Exception, match="ctk_p2p_alloc"):
        e.p2p_step(s)
    h = e.p2p_alloc(0, 1)
    assert len(h) == 64
    e.p2p_connect([h])
    for t in range(3):
        np.testing.assert_allclose(e.p2p_step(s), f.step(s), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(e.read("U_NOM"), f.read("U_NOM"), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        e.p2p_alloc(3, 2)
    c = CtkEngine("cem", "ODE", num_rollouts=64, mpc_horizon=10, dt=0.02, cem_outer_it=1, cem_best_k=8)
    with pytest.raises(Exception, match="not MPPI"):
        c.p2p_alloc(0, 1)
    for x in (e, f, c):
        x.close()
    # the network predictor: N <= 8192 runs the pair form of the rollout kernel, whose exchanging instantiation is its own
    from oracle import ctk_oracle as O
    w = O.mlp_default_weights(4)
    e = CtkEngine("mppi", "MLP", num_rollouts=1024, mpc_horizon=15, dt=0.02, seed=3, period_interpolation_inducing_points=5)
    f = CtkEngine("mppi", "MLP", num_rollouts=1024, mpc_horizon=15, dt=0.02, seed=3, period_interpolation_inducing_points=5)
    e.set_predictor_weights(w); f.set_predictor_weights(w)
    e.p2p_connect([e.p2p_alloc(0, 1)])
    for t in range(3):
        np.testing.assert_allclose(e.p2p_step(s), f.step(s), rtol=1e-5, atol=1e-6)
    e.close(); f.close()


def test_set_stream_orders_by_event_and_get_stream_reports_it():
    """ctk_set_stream used to synchronise the host on every rebind; now the handle's earlier work is ordered before the new
    stream by an event.  A handle hopping between two torch streams every step computes what a handle on its own stream computes."""
    import numpy as np
    import torch
    from control_toolkit_amd import CtkEngine
    kw = dict(num_rollouts=512, mpc_horizon=30, dt=0.02, period_interpolation_inducing_points=5, seed=9)
    a, b = CtkEngine("mppi", "ODE", **kw), CtkEngine("mppi", "ODE", **kw)
    own = b.get_stream()
    assert own != 0
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    for t in range(8):
        st = streams[t & 1]
        b.set_stream(st.cuda_stream)
        assert b.get_stream() == st.cuda_stream
        b.set_stream(st.cuda_stream)                                  # rebinding to the same stream is a no-op
        np.testing.assert_array_equal(b.step(s), a.step(s))
        s = s + np.array([0.01, 0.0, -0.02, 0.01], np.float32)
    np.testing.assert_array_equal(b.read("U_NOM"), a.read("U_NOM"))
    b.resident_enable(True, 200.0)                                    # a handle on a caller's stream keeps it (include/ctk_hip.h)
    assert b.get_stream() == streams[1].cuda_stream
    a.resident_enable(True, 200.0)                                    # ... a handle on its own stream moves to a high-priority one
    np.testing.assert_array_equal(b.step(s), a.step(s))
    assert a.get_stream() != 0
    a.close(); b.close()
