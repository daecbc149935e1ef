"""CPU, world_size = 2, gloo: the sharded-MPPI collective plumbing (control_toolkit_amd/dist.py).
The engine is replaced by a tiny stand-in built on the oracle (test infrastructure) that writes
the same (2+P)-float record libctk_hip.so writes; the test checks that one all-gather + replicated
merge reproduces the single-process result on both ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardEngine:
    """Implements the engine surface ShardedMPPI uses (mppi_partial_size / step_begin / step_end)
    with oracle math on CPU tensors addressed by data_ptr, like the C ABI does on the GPU."""

    def __init__(self, N_local, H, p, rank):
        from oracle import ctk_oracle as O
        self.O = O
        pred = O.Predictor("ODE")
        self.m = O.MPPI(pred, O.Cost(pred.env), num_rollouts=N_local, mpc_horizon=H, period_interpolation_inducing_points=p)
        self.rank = rank
        self._views = {}

    def mppi_partial_size(self):
        return 2 + self.m.P

    def register(self, tensor):
        self._views[tensor.data_ptr()] = tensor

    def mppi_step_begin(self, s, partial_ptr, samples=None, u_prev=None):
        O, m = self.O, self.m
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, 4), (m.N, 1))
        self.u_nom_shift = np.concatenate([m.u_nom[:, 1:, :], m.u_nom[:, -1:, :]], 1)
        tile = np.asarray(samples, np.float32) * m.stdev
        du = O.interpolate(tile, m.M)
        u_run = np.clip(np.tile(self.u_nom_shift, (m.N, 1, 1)) + du, m.low, m.high)
        traj = m.predictor.predict_core(s_t, u_run)
        J = m.cost.get_trajectory_cost(traj, u_run, np.array([m.u], np.float32)) + m.mppi_correction_cost(u_run, du)
        rho, a, b = m.mppi_partials(J, tile)
        rec = np.concatenate([[rho, a], b.reshape(-1)]).astype(np.float32)
        self._views[partial_ptr].copy_(torch.from_numpy(rec))

    def mppi_step_end(self, parts_ptr, n_parts):
        O, m = self.O, self.m
        parts = self._views[parts_ptr].numpy().reshape(n_parts, -1)
        _, _, b = O.merge_mppi_partials(parts[:, 0], parts[:, 1], parts[:, 2:, None], m.LBD)
        w = O.interpolate(b.reshape(1, m.P, 1), m.M)
        m.u_nom = np.clip(self.u_nom_shift + w, m.low, m.high).astype(np.float32)
        m.u = np.float32(m.u_nom[0, 0, 0])
        return np.array([m.u], np.float32)


def _worker(rank, world, port, N, H, p, noise, s, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from control_toolkit_amd.dist import ShardedMPPI
    Nl = N // world
    eng = OracleShardEngine(Nl, H, p, rank)
    sh = ShardedMPPI(eng, rank, world)
    eng.register(sh.mine); eng.register(sh.all)
    us = []
    for t in range(noise.shape[0]):
        us.append(float(sh.step(s, noise[t, rank * Nl:(rank + 1) * Nl])[0]))
    out_q.put((rank, us, eng.m.u_nom.copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_mppi_two_ranks_gloo():
    sys.path.insert(0, ROOT)
    from oracle import ctk_oracle as O
    N, H, p, steps = 128, 20, 5, 3
    rng = np.random.default_rng(0)
    P = O.num_inducing_points(H, p)
    noise = rng.standard_normal((steps, N, P, 1)).astype(np.float32)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    pred = O.Predictor("ODE")
    full = O.MPPI(pred, O.Cost(pred.env), num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=p)
    ref = [float(full.step(s, noise[t])) for t in range(steps)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, H, p, noise, s, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=100) for _ in procs]
    for pr in procs:
        pr.join(timeout=30)
        assert pr.exitcode == 0
    res.sort(key=lambda r: r[0])
    np.testing.assert_array_equal(res[0][1], res[1][1])             # identical on both ranks, no 2nd collective
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(res[0][2], full.u_nom, rtol=1e-4, atol=2e-5)


class FakeP2PEngine:
    """Engine surface ShardedMPPI's p2p set-up touches, with a failure injected on ONE rank: whatever happens
    locally, both ranks must run the same sequence of collectives and end on the same exchange."""

    def __init__(self, rank, mode):
        self.rank, self.mode, self.plan, self.closed = rank, mode, 0.0, False

    def mppi_partial_size(self): return 4
    def samples_needed(self): return 8
    def get_state(self): return np.zeros(3, np.float32)
    def set_state(self, st): self.plan = 0.0
    def read(self, name): return np.full((1, 3, 1), self.plan, np.float32)

    def p2p_alloc(self, rank, world):
        if self.mode == "alloc_fail" and rank == 1:
            raise RuntimeError("hipIpcGetMemHandle: invalid argument")
        return bytes(64)

    def p2p_connect(self, handles):
        assert len(handles) == 2 and all(len(h) == 64 for h in handles)
        if self.mode == "connect_fail" and self.rank == 0:
            raise RuntimeError("hipIpcOpenMemHandle: invalid device pointer")

    def p2p_step(self, s, samples=None, u_prev=None):
        if self.mode == "step_fail" and self.rank == 1:
            raise RuntimeError("ctk_p2p_step: timed out waiting for a peer's record")
        self.plan = 2.0 if (self.mode == "differs" and self.rank == 0) else 1.0
        return np.array([self.plan], np.float32)

    def p2p_close(self): self.closed = True
    def mppi_step_begin(self, s, ptr, samples=None, u_prev=None): pass

    def mppi_step_end(self, ptr, n):
        self.plan = 1.0
        return np.array([1.0], np.float32)


def _p2p_worker(rank, world, port, modes, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from control_toolkit_amd.dist import ShardedMPPI
    res = {}
    for mode in modes:
        eng = FakeP2PEngine(rank, mode)
        sh = ShardedMPPI(eng, rank, world, exchange="p2p")
        u = sh.step(np.zeros(4, np.float32))                 # the step after the set-up still lines up across ranks
        dist.barrier()
        res[mode] = (sh.exchange, sh.p2p_error, eng.closed, float(u[0]))
    out_q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_p2p_setup_falls_back_collectively():
    modes = ["ok", "alloc_fail", "connect_fail", "step_fail", "differs"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 991) % 2000)
    procs = [ctx.Process(target=_p2p_worker, args=(r, 2, port, modes, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for pr in procs:
        pr.join(timeout=30)
        assert pr.exitcode == 0
    for mode in modes:
        ex0, err0, closed0, u0 = res[0][mode]
        ex1, err1, closed1, u1 = res[1][mode]
        assert ex0 == ex1 == ("p2p" if mode == "ok" else "rccl"), (mode, res[0][mode], res[1][mode])
        assert u0 == u1 == 1.0
        if mode != "ok":
            assert closed0 and closed1 and (err0 or err1)
    with pytest.raises(ValueError):
        from control_toolkit_amd.dist import ShardedMPPI
        ShardedMPPI(FakeP2PEngine(0, "ok"), 0, 1, exchange="smoke-signals")
