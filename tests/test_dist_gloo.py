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


# ------------------------------------------------------------------------------------------------------------------
# ShardedTopK (CEM / random-action) and ShardedRPGD: oracle-backed stand-in engines that write the SAME records
# libctk_hip.so writes (include/ctk_hip.h: ctk_shard_* / ctk_rpgd_step_*); world size 2 over gloo must reproduce the
# single-process oracle on the whole population.
# ------------------------------------------------------------------------------------------------------------------
class OracleTopKEngine:
    """records {J, global index (int bits), Q[H]}, best K of the shard, sorted by (J, index)"""

    def __init__(self, kind, N_local, H, K, rank, cem_outer_it=3):
        from oracle import ctk_oracle as O
        self.O, self.kind, self.N, self.H, self.rank = O, kind, N_local, H, rank
        pred = O.Predictor("ODE")
        self.pred, self.cost = pred, O.Cost(pred.env)
        self.K = K if kind == "cem" else 1
        self.its = cem_outer_it if kind == "cem" else 1
        self.low, self.high = np.float32(-1.0), np.float32(1.0)
        self.mu = np.zeros((1, H, 1), np.float32)
        self.std = np.full((1, H, 1), 0.5, np.float32)
        self.u = np.float32(0.0)
        self._views = {}

    def register(self, t):
        self._views[t.data_ptr()] = t

    def shard_candidates_size(self): return self.K * (2 + self.H)
    def shard_iterations(self): return self.its

    def shard_iter_begin(self, s, cand_ptr, samples=None, u_prev=None):
        O = self.O
        s_t = np.tile(np.asarray(s, np.float32).reshape(1, 4), (self.N, 1))
        smp = np.asarray(samples, np.float32)
        if self.kind == "cem":
            Q = np.clip(np.tile(self.mu, (self.N, 1, 1)) + smp * self.std, self.low, self.high)          # optimizer_cem_tf.py:64-66
        else:
            Q = (smp * (self.high - self.low) + self.low).astype(np.float32)                              # optimizer_random_action_tf.py:56-61
        J = self.cost.get_trajectory_cost(self.pred.predict_core(s_t, Q), Q, np.array([self.u], np.float32))
        best = O.argsort_total_order(J)[: self.K]
        rec = np.zeros((self.K, 2 + self.H), np.float32)
        rec[:, 0] = J[best]
        rec[:, 1] = (best + self.rank * self.N).astype(np.int32).view(np.float32)
        rec[:, 2:] = Q[best, :, 0]
        self._views[cand_ptr].copy_(torch.from_numpy(rec.ravel()))

    def shard_iter_end(self, cands_ptr, n_ranks):
        rec = self._views[cands_ptr].numpy().reshape(n_ranks * self.K, 2 + self.H)
        best = self.O.argsort_total_order(rec[:, 0])[: self.K]      # positional ties == global-index ties (rank-major, sorted lists)
        self.elite = rec[best, 2:][:, :, None].copy()
        self.best_global = rec[best, 1].copy().view(np.int32)
        if self.kind == "cem":
            self.mu = np.mean(self.elite, axis=0, keepdims=True, dtype=np.float32)
            self.std = np.sqrt(np.mean((self.elite - self.mu) ** 2, axis=0, keepdims=True, dtype=np.float32)).astype(np.float32)

    def shard_finish(self):
        if self.kind == "cem":                                       # optimizer_cem_tf.py:99-102
            self.std = np.clip(self.std, np.float32(0.01), np.float32(1e8))
            self.std = np.concatenate([self.std[:, 1:], np.full((1, 1, 1), 0.5, np.float32)], 1)
            self.mu = np.concatenate([self.mu[:, 1:], np.zeros((1, 1, 1), np.float32)], 1)
        self.u = np.float32(self.elite[0, 0, 0])
        return np.array([self.u], np.float32)


def _topk_worker(rank, world, port, kind, N, H, K, noise, s, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from control_toolkit_amd.dist import ShardedTopK
    Nl = N // world
    eng = OracleTopKEngine(kind, Nl, H, K, rank, cem_outer_it=noise.shape[1])
    sh = ShardedTopK(eng, rank, world)
    eng.register(sh.mine); eng.register(sh.all)
    us = [float(sh.step(s, noise[t][:, rank * Nl:(rank + 1) * Nl])[0]) for t in range(noise.shape[0])]
    out_q.put((rank, us, eng.mu.copy(), eng.std.copy(), eng.best_global.copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("kind", ["cem", "random_action"])
def test_sharded_topk_two_ranks_gloo(kind):
    sys.path.insert(0, ROOT)
    from oracle import ctk_oracle as O
    N, H, K, steps, its = 96, 12, 10, 3, (3 if kind == "cem" else 1)
    rng = np.random.default_rng(5)
    noise = (rng.standard_normal((steps, its, N, H, 1)) if kind == "cem" else rng.random((steps, its, N, H, 1))).astype(np.float32)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    pred = O.Predictor("ODE")
    if kind == "cem":
        full = O.CEM(pred, O.Cost(pred.env), num_rollouts=N, mpc_horizon=H, cem_outer_it=its, cem_best_k=K)
        ref = [float(full.step(s, noise[t])) for t in range(steps)]
    else:
        full = O.RandomAction(pred, O.Cost(pred.env), num_rollouts=N, mpc_horizon=H)
        ref = [float(full.step(s, noise[t, 0])) for t in range(steps)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + (313 if kind == "cem" else 717)) % 2000)
    procs = [ctx.Process(target=_topk_worker, args=(r, 2, port, kind, N, H, K, noise, s, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=150) for _ in procs], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=30)
        assert pr.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])          # replicated selection: identical on both ranks
    np.testing.assert_array_equal(res[0][4], res[1][4])
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-6, atol=1e-7)
    if kind == "cem":
        np.testing.assert_allclose(res[0][2], full.dist_mue, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(res[0][3], full.stdev, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(res[0][4], full.best_idx)  # global indices of the last iteration's elites
    else:
        assert int(res[0][4][0]) == int(full.best_idx)


class OracleRpgdShardEngine:
    """keeper records {J, global index, age, Q[H], m[H], v[H]} of the shard's best min(k, N_local) plans, sorted"""

    def __init__(self, N_local, world, H, p, k_global, rank, outer_its, resamp_per):
        from oracle import ctk_oracle as O
        self.O, self.N, self.H, self.rank, self.world = O, N_local, H, rank, world
        pred = O.Predictor("ODE")
        self.r = O.RPGD(pred, O.Cost(pred.env), num_rollouts=N_local, mpc_horizon=H, outer_its=outer_its, resamp_per=resamp_per,
                        period_interpolation_inducing_points=p, opt_keep_k_ratio=1.0)
        self.k = k_global
        self.kl = min(k_global, N_local)
        self._views = {}

    def register(self, t):
        self._views[t.data_ptr()] = t

    def reset(self, draws):
        self.r.optimizer_reset(draws)

    def rpgd_keepers_size(self): return self.kl * (3 + 3 * self.H)

    def rpgd_fresh_rows(self, n_ranks):
        if self.r.count % self.r.resamp_per != 0:
            return 0
        first_keeper = n_ranks * self.N - self.k
        return int(np.clip(first_keeper - self.rank * self.N, 0, self.N))

    def rpgd_step_begin(self, s, keep_ptr, u_prev=None):
        O, r = self.O, self.r
        self.s_t = np.tile(np.asarray(s, np.float32).reshape(1, 4), (self.N, 1))
        for _ in range(r.first_iter_count if r.count == 0 else r.outer_its):
            r.grad_step(self.s_t)
        traj = r.predictor.predict_core(self.s_t, r.Q)
        J = r.cost.get_trajectory_cost(traj, r.Q, np.array([r.u], np.float32))
        best = O.argsort_total_order(J)[: self.kl]
        H = self.H
        rec = np.zeros((self.kl, 3 + 3 * H), np.float32)
        rec[:, 0] = J[best]
        rec[:, 1] = (best + self.rank * self.N).astype(np.int32).view(np.float32)
        rec[:, 2] = r.trajectory_ages[best]
        rec[:, 3:3 + H] = r.Q[best, :, 0]
        rec[:, 3 + H:3 + 2 * H] = r.opt.m[best, :, 0]
        rec[:, 3 + 2 * H:] = r.opt.v[best, :, 0]
        self._views[keep_ptr].copy_(torch.from_numpy(rec.ravel()))

    def rpgd_step_end(self, keep_all_ptr, n_ranks, draws=None):
        O, r, H, N = self.O, self.r, self.H, self.N
        rec = self._views[keep_all_ptr].numpy().reshape(n_ranks * self.kl, 3 + 3 * H)
        keep = O.argsort_total_order(rec[:, 0])[: self.k]          # global keep-k, sorted
        sp = r.shift_previous
        shiftq = lambda q: np.concatenate([q[:, sp:], np.tile(q[:, -1:], (1, sp))], 1)
        shift1 = lambda a: np.concatenate([a[:, 1:], np.zeros((a.shape[0], 1), np.float32)], 1)
        u_nom = rec[keep[0], 3:3 + H].copy()
        if r.count % r.resamp_per == 0:
            first_keeper = n_ranks * N - self.k
            Q = np.zeros((N, H), np.float32); m = np.zeros((N, H), np.float32); v = np.zeros((N, H), np.float32)
            ages = np.zeros((N,), np.float32)
            n_fresh = self.rpgd_fresh_rows(n_ranks)
            if n_fresh:
                Q[:n_fresh] = r.sample_actions(draws)[:, :, 0]
            for i in range(n_fresh, N):
                kk = keep[self.rank * N + i - first_keeper]
                Q[i] = shiftq(rec[kk:kk + 1, 3:3 + H])[0]
                m[i] = shift1(rec[kk:kk + 1, 3 + H:3 + 2 * H])[0]
                v[i] = shift1(rec[kk:kk + 1, 3 + 2 * H:])[0]
                ages[i] = rec[kk, 2]
            r.Q, r.opt.m, r.opt.v, r.trajectory_ages = Q[:, :, None], m[:, :, None], v[:, :, None], ages
        else:
            r.Q = shiftq(r.Q[:, :, 0])[:, :, None]
            r.opt.m, r.opt.v = shift1(r.opt.m[:, :, 0])[:, :, None], shift1(r.opt.v[:, :, 0])[:, :, None]
        r.trajectory_ages = r.trajectory_ages + np.float32(1.0)
        r.count += 1
        r.u = np.float32(u_nom[0])
        return np.array([r.u], np.float32)


def _rpgd_worker(rank, world, port, N, H, p, k, outer_its, resamp_per, reset_draws, fresh, s_seq, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from control_toolkit_amd.dist import ShardedRPGD
    Nl = N // world
    eng = OracleRpgdShardEngine(Nl, world, H, p, k, rank, outer_its, resamp_per)
    eng.reset(reset_draws[rank * Nl:(rank + 1) * Nl])
    sh = ShardedRPGD(eng, rank, world)
    eng.register(sh.mine); eng.register(sh.all)
    us = []
    for t, s in enumerate(s_seq):
        nf = sh.fresh_rows()
        d = None
        if nf:                                    # this shard's fresh rows are global rows [rank*Nl, rank*Nl + nf)
            d = fresh[t][rank * Nl: rank * Nl + nf]
        us.append(float(sh.step(s, d)[0]))
    out_q.put((rank, us, eng.r.Q.copy(), eng.r.opt.m.copy(), eng.r.opt.v.copy(), eng.r.trajectory_ages.copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_sharded_rpgd_two_ranks_gloo():
    sys.path.insert(0, ROOT)
    from oracle import ctk_oracle as O
    N, H, p, k, outer_its, resamp_per, steps = 32, 12, 4, 6, 2, 2, 4
    rng = np.random.default_rng(11)
    P = O.num_inducing_points(H, p)
    reset_draws = rng.random((N, P, 1), dtype=np.float32)
    fresh = [rng.random((N - k, P, 1), dtype=np.float32) for _ in range(steps)]
    s_seq = [np.array([0.05 + 0.01 * t, -0.1, 2.8 - 0.02 * t, 0.4], np.float32) for t in range(steps)]
    pred = O.Predictor("ODE")
    full = O.RPGD(pred, O.Cost(pred.env), num_rollouts=N, mpc_horizon=H, outer_its=outer_its, resamp_per=resamp_per,
                  period_interpolation_inducing_points=p, opt_keep_k_ratio=k / N)
    assert full.k == k
    full.optimizer_reset(reset_draws)
    ref = [float(full.step(s_seq[t], fresh[t])) for t in range(steps)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 1201) % 2000)
    procs = [ctx.Process(target=_rpgd_worker, args=(r, 2, port, N, H, p, k, outer_its, resamp_per, reset_draws, fresh, s_seq, q))
             for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=200) for _ in procs], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=30)
        assert pr.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-6, atol=1e-7)
    for j, ref_arr in ((2, full.Q), (3, full.opt.m), (4, full.opt.v)):
        got = np.concatenate([res[0][j], res[1][j]], 0)        # the two shards ARE the global population, row for row
        np.testing.assert_allclose(got, ref_arr, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(np.concatenate([res[0][5], res[1][5]]), full.trajectory_ages)
