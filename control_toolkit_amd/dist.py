"""Sharded MPPI over the GPUs of one node: one process per GPU, rollouts
[rank*N_local, (rank+1)*N_local) per rank, ONE all-gather of the (2+P)-float soft-min record per
step over RCCL (torch.distributed backend "nccl" is RCCL on ROCm), then every rank merges the
records with the same kernel and arrives at the identical u_nom without a second collective.
The payload is 408 B per rank at cfg5, so the collective is latency-bound, not xGMI-bandwidth-bound.

The engine is injected so that the collective plumbing can be exercised on CPU with the gloo
backend (tests/test_dist_gloo.py); on the GPU box the engine is a CtkEngine."""
from __future__ import annotations

import numpy as np


class ShardedMPPI:
    def __init__(self, engine, rank: int, world_size: int, group=None, device=None, always_collective: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine, self.rank, self.world_size, self.group = engine, rank, world_size, group
        self.always_collective = always_collective   # issue the all-gather even for one rank (rehearsal of the RCCL path)
        self.rec = int(engine.mppi_partial_size())
        self.device = device if device is not None else torch.device("cpu")
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.all = torch.zeros(self.rec * world_size, dtype=torch.float32, device=self.device)

    def step(self, s, samples=None, u_prev=None) -> np.ndarray:
        """samples: this rank's slice of the draws (host array / device pointer) or None (device
        Philox addressed by GLOBAL rollout index, so the result does not depend on world_size)."""
        self.engine.mppi_step_begin(s, self.mine.data_ptr(), samples, u_prev=u_prev)
        if self.world_size > 1 or self.always_collective:
            self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
            parts = self.all
        else:
            parts = self.mine
        return self.engine.mppi_step_end(parts.data_ptr(), self.world_size)


class ShardedTopK:
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
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.all = torch.zeros(self.rec * world_size, dtype=torch.float32, device=self.device)

    def step(self, s, samples=None, u_prev=None) -> np.ndarray:
        """samples: None (device Philox by global rollout index) or this rank's draws [iterations, N_local, H, 1]."""
        its = self.engine.shard_iterations()
        for it in range(its):
            smp = None if samples is None else samples[it]
            self.engine.shard_iter_begin(s, self.mine.data_ptr(), smp, u_prev=u_prev)
            if self.world_size > 1:
                self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
                parts = self.all
            else:
                parts = self.mine
            self.engine.shard_iter_end(parts.data_ptr(), self.world_size)
        return self.engine.shard_finish()


class ShardedRPGD:
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
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.all = torch.zeros(self.rec * world_size, dtype=torch.float32, device=self.device)

    def fresh_rows(self) -> int:
        return self.engine.rpgd_fresh_rows(self.world_size)

    def step(self, s, draws=None, u_prev=None) -> np.ndarray:
        """draws: None (device Philox by global row) or raw draws [fresh_rows(), P, 1] for this shard."""
        self.engine.rpgd_step_begin(s, self.mine.data_ptr(), u_prev=u_prev)
        if self.world_size > 1:
            self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
            parts = self.all
        else:
            parts = self.mine
        return self.engine.rpgd_step_end(parts.data_ptr(), self.world_size, draws)
