"""Sharded MPPI over the GPUs of one node: one process per GPU, rollouts
[rank*N_local, (rank+1)*N_local) per rank, ONE all-gather of the (2+P)-float soft-min record per
step over RCCL (torch.distributed backend "nccl" is RCCL on ROCm), then every rank merges the
records with the same kernel and arrives at the identical u_nom without a second collective.
The payload is 408 B per rank at cfg5, so the collective is latency-bound, not xGMI-bandwidth-bound.

The engine is injected so that the collective plumbing can be exercised on CPU with the gloo
backend (tests/test_dist_gloo.py); on the GPU box the engine is a CtkEngine.

What does and does not depend on the world size: the DRAWS do not (device Philox is addressed by the global rollout index),
and neither does the algorithm (the merged record / the global best-K / the global keep-k are those of the whole
population).  The BITS of a result agree across world sizes to fp32 rounding only: partial sums are associated per shard,
and the MLP rollout kernel's form depends on the shard size (<= 8192 rollouts: two waves per tile, larger: one wave per
tile — another summation order inside the network step).  tests/test_gpu_mlp.py holds 8 shards of 8192 against one handle
of 65536 at rtol 3e-6 on the costs."""
from __future__ import annotations

import numpy as np


def _bind_stream(engine, torch, device):
    """The collective runs on torch's current stream of `device`; the engine's begin/end kernels read and write
    the exchanged records, so they must be ordered with it: put the engine on the same stream (include/ctk_hip.h:
    ctk_set_stream).  CPU tensors (gloo tests with a stand-in engine): nothing to order."""
    if device is not None and getattr(device, "type", "cpu") == "cuda" and hasattr(engine, "set_stream"):
        handle = torch.cuda.current_stream(device).cuda_stream
        engine.set_stream(handle)
        return handle
    return None


class _StreamBound:
    """begin -> collective -> end must run on ONE stream (they hand the `mine` / `all` record buffers to each other).  The
    collective runs on whatever torch stream is current when step() is called, so step() re-binds the engine whenever
    that stream is not the one bound last (a caller stepping under `with torch.cuda.stream(...)`): ctk_set_stream orders
    the engine's earlier work before the new stream's (include/ctk_hip.h)."""

    _bound = None

    def _bind(self):
        self._bound = _bind_stream(self.engine, self.torch, self.device)

    def _rebind_if_stream_changed(self):
        if self._bound is None:
            return
        if self.torch.cuda.current_stream(self.device).cuda_stream != self._bound:
            self._bind()

    # ---- where a sharded step's time goes (bench.py: exchange_us) ---------------------------------------------------------
    # With timing on, every step leaves four events on the stream: start | begin kernels done | exchange done | end kernels done.
    # exchange = the collective as the stream sees it (its own time + waiting for the slowest peer); off by default: recording
    # events costs host time.
    _timing = None

    def enable_timing(self, on: bool = True):
        cuda = getattr(self.device, "type", "cpu") == "cuda"
        self._timing = [] if (on and cuda) else None

    def _mark(self, marks):
        if marks is not None:
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record(self.torch.cuda.current_stream(self.device))
            marks.append(ev)

    def _new_marks(self):
        if self._timing is None:
            return None
        marks = []
        self._timing.append(marks)
        self._mark(marks)
        return marks

    def timing_us(self):
        """per recorded step: (begin_us, exchange_us, end_us) lists — device-timeline intervals between the four events; a step with
        several exchanges (CEM: one per outer iteration) contributes their sums.  Call after a device synchronize."""
        out = {"begin_us": [], "exchange_us": [], "end_us": []}
        for marks in self._timing or []:
            b = x = e = 0.0
            for i in range(0, len(marks) - 3, 3):
                b += marks[i].elapsed_time(marks[i + 1]); x += marks[i + 1].elapsed_time(marks[i + 2]); e += marks[i + 2].elapsed_time(marks[i + 3])
            out["begin_us"].append(b * 1e3); out["exchange_us"].append(x * 1e3); out["end_us"].append(e * 1e3)
        return out


class _EngineSnapshot:
    """Everything a step changes, for the p2p self-test's rollback: the warm-start state (ctk_get_state), the
    recurrent predictor's carried hidden state and the Philox position."""

    def __init__(self, engine):
        self.engine = engine
        self.state = engine.get_state()
        self.hidden = engine.predictor_get_hidden() if getattr(engine, "predictor_hidden_size", lambda: 0)() else None
        self.call = engine.rng_position() if hasattr(engine, "rng_position") else None

    def restore(self):
        self.engine.set_state(self.state)
        if self.hidden is not None:
            self.engine.predictor_set_hidden(self.hidden)
        if self.call is not None:
            self.engine.set_rng_position(self.call)


class ShardedMPPI(_StreamBound):
    """exchange = "rccl": ctk_mppi_step_begin -> all_gather_into_tensor -> ctk_mppi_step_end.
    exchange = "p2p" : the ranks' exchange kernels store their records straight into each other's HBM over xGMI
    (ctk_p2p_*, HIP IPC mappings set up once through the process group): no collective launch and no host
    round trip inside a step.  One node only.  If the set-up or the self-test steps fail on any rank, all
    ranks fall back to "rccl" together (the decision is itself all-reduced)."""

    SELF_TEST_STEPS = 16

    def __init__(self, engine, rank: int, world_size: int, group=None, device=None, always_collective: bool = False,
                 exchange: str = "rccl"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine, self.rank, self.world_size, self.group = engine, rank, world_size, group
        self.always_collective = always_collective   # issue the all-gather even for one rank (rehearsal of the RCCL path)
        self.rec = int(engine.mppi_partial_size())
        self.device = device if device is not None else torch.device("cpu")
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.all = torch.zeros(self.rec * world_size, dtype=torch.float32, device=self.device)
        self.exchange = "rccl"
        self.p2p_error = None
        self._bind()
        if exchange == "p2p" and world_size > 1:
            self._try_p2p(device)
        elif exchange not in ("rccl", "p2p"):
            raise ValueError(f"exchange must be 'rccl' or 'p2p', got {exchange!r}")

    def _all_ok(self, ok: bool, device) -> bool:
        dist, torch = self.dist, self.torch
        backend = dist.get_backend(self.group)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _try_p2p(self, device):
        dist = self.dist
        ok, handle = True, b"\0" * 64
        try:
            handle = self.engine.p2p_alloc(self.rank, self.world_size)
        except Exception as e:   # noqa: BLE001 — any failure means "use RCCL", decided collectively below
            ok, self.p2p_error = False, f"alloc: {e}"
        handles = [None] * self.world_size
        dist.all_gather_object(handles, handle, group=self.group)
        if ok:
            try:
                self.engine.p2p_connect(handles)
            except Exception as e:   # noqa: BLE001
                ok, self.p2p_error = False, f"connect: {e}"
        ok = self._all_ok(ok, device)
        if ok:
            # self-test: two exchange steps (both parities) must complete on every rank and reproduce the plan of
            # the RCCL path.  Every rank runs the same sequence of collectives whatever happens locally.
            snap = _EngineSnapshot(self.engine)
            s0 = np.zeros(int(getattr(self.engine, "S", 4)), np.float32)
            # explicit draws (the device sampler advances with every step, the two paths would see different noise)
            noise = np.random.default_rng(1234 + self.rank).standard_normal(self.engine.samples_needed()).astype(np.float32)
            u_p2p = plan_p2p = None
            dist.barrier(group=self.group)
            try:
                for _ in range(self.SELF_TEST_STEPS):       # both buffer halves, many times
                    u_p2p = self.engine.p2p_step(s0, noise)
                plan_p2p = self.engine.read("U_NOM")
            except Exception as e:   # noqa: BLE001
                ok, self.p2p_error = False, f"self-test: {e}"
            ok = self._all_ok(ok, device)
            snap.restore()
            if ok:
                for _ in range(self.SELF_TEST_STEPS):       # exchange == "rccl" here
                    u_rccl = self.step(s0, noise)
                if not (np.allclose(u_p2p, u_rccl, rtol=1e-4, atol=2e-5)
                        and np.allclose(plan_p2p, self.engine.read("U_NOM"), rtol=1e-4, atol=2e-5)):
                    ok, self.p2p_error = False, "self-test: p2p and rccl plans differ"
                snap.restore()
                ok = self._all_ok(ok, device)
        self.exchange = "p2p" if ok else "rccl"
        if not ok:
            try:
                self.engine.p2p_close()
            except Exception:   # noqa: BLE001
                pass

    def step(self, s, samples=None, u_prev=None) -> np.ndarray:
        """samples: this rank's slice of the draws (host array / device pointer) or None (device
        Philox addressed by GLOBAL rollout index, so the draws do not depend on world_size; the result agrees across
        world sizes to fp32 rounding — see the module docstring)."""
        self._rebind_if_stream_changed()
        if self.exchange == "p2p":
            return self.engine.p2p_step(s, samples, u_prev=u_prev)
        marks = self._new_marks()
        self.engine.mppi_step_begin(s, self.mine.data_ptr(), samples, u_prev=u_prev)
        self._mark(marks)
        if self.world_size > 1 or self.always_collective:
            self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
            parts = self.all
        else:
            parts = self.mine
        self._mark(marks)
        if marks is None:
            return self.engine.mppi_step_end(parts.data_ptr(), self.world_size)
        # timed pass: the end launch is issued, marked, and only then waited for (the result is published by the end kernel)
        u = self.engine.mppi_step_end(parts.data_ptr(), self.world_size)
        self._mark(marks)
        return u


class ShardedTopK(_StreamBound):
    """Sharded CEM / random-action: per outer iteration every rank rolls out its shard
    and contributes its best K plans as records {J, global index, Q[H]}; ONE all-gather of those records
    per iteration; every rank then selects the global best K from the union and (CEM) refits the same
    mean / stdev — no second collective.  cfg3: K*(2+H) = 409*32 floats = 51 KiB per rank."""

    def __init__(self, engine, rank: int, world_size: int, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine, self.rank, self.world_size, self.group = engine, rank, world_size, group
        self.rec = int(engine.shard_candidates_size())
        self.device = device if device is not None else torch.device("cpu")
        self._bind()
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.all = torch.zeros(self.rec * world_size, dtype=torch.float32, device=self.device)

    def step(self, s, samples=None, u_prev=None) -> np.ndarray:
        """samples: None (device Philox by global rollout index) or this rank's draws [iterations, N_local, H, 1]."""
        self._rebind_if_stream_changed()
        its = self.engine.shard_iterations()
        marks = self._new_marks()
        for it in range(its):
            smp = None if samples is None else samples[it]
            self.engine.shard_iter_begin(s, self.mine.data_ptr(), smp, u_prev=u_prev)
            self._mark(marks)
            if self.world_size > 1:
                self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
                parts = self.all
            else:
                parts = self.mine
            self._mark(marks)
            self.engine.shard_iter_end(parts.data_ptr(), self.world_size)
            self._mark(marks)
        return self.engine.shard_finish()


class ShardedRPGD(_StreamBound):
    """Sharded RPGD: the Adam descent is local; one all-gather per step of the shards'
    best plans WITH their optimizer state, after which every rank rebuilds its rows of the global
    population [fresh | keepers sorted by cost] exactly as one big optimizer would."""

    def __init__(self, engine, rank: int, world_size: int, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine, self.rank, self.world_size, self.group = engine, rank, world_size, group
        self.rec = int(engine.rpgd_keepers_size())
        self.device = device if device is not None else torch.device("cpu")
        self._bind()
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.all = torch.zeros(self.rec * world_size, dtype=torch.float32, device=self.device)

    def fresh_rows(self) -> int:
        return self.engine.rpgd_fresh_rows(self.world_size)

    def step(self, s, draws=None, u_prev=None) -> np.ndarray:
        """draws: None (device Philox by global row) or raw draws [fresh_rows(), P, 1] for this shard."""
        self._rebind_if_stream_changed()
        marks = self._new_marks()
        self.engine.rpgd_step_begin(s, self.mine.data_ptr(), u_prev=u_prev)
        self._mark(marks)
        if self.world_size > 1:
            self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
            parts = self.all
        else:
            parts = self.mine
        self._mark(marks)
        u = self.engine.rpgd_step_end(parts.data_ptr(), self.world_size, draws)
        self._mark(marks)
        return u
