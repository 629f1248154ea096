"""Data parallelism of the MAPPO update: one process per GPU, environments sharded by rank, replicated
parameters.  Rollout, trajectory writing and GAE need no communication; the update has exactly two exchange
points:

* once per update, SUM of the 3 float64 advantage moments {count, sum, sum of squares}: the reference
  normalises advantages over the WHOLE buffer (mappo_algorithm.py:293);
* once per SGD iteration, SUM of one flat float32 buffer = [actor grads | critic grads | 2x4 loss sums].
  Every rank's gradient kernels already scale by 1/L_global ("gradient of sums"), so the sum over ranks IS the
  full-batch gradient; clip + Adam then run identically on every rank.

``torch.distributed`` with backend "nccl" is RCCL on ROCm.  At 76 KB the gradient all-reduce is latency bound;
one fused buffer keeps it to a single collective per iteration."""
from __future__ import annotations

import torch


class GradSync:
    def __init__(self, process_group=None, enabled=None, always=None):
        """``always``: issue the collectives even in a one-rank group (default: environment variable
        EDGEMARL_SYNC_ALWAYS=1).  A one-GPU box can then execute the production backend's call path (RCCL all-reduce
        ordered against the kernels this library launches on torch's current stream); results are unchanged."""
        import os

        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        self.enabled = (dist.is_available() and dist.is_initialized()) if enabled is None else bool(enabled)
        self.world_size = dist.get_world_size(process_group) if self.enabled else 1
        self.rank = dist.get_rank(process_group) if self.enabled else 0
        if always is None:
            always = os.environ.get("EDGEMARL_SYNC_ALWAYS", "0") == "1"
        self.active = self.enabled and (self.world_size > 1 or bool(always))
        # gloo (the CPU test rig; production is "nccl" = RCCL) stages device tensors through the host from its own
        # threads: with kernels still in flight on the caller's stream that was seen to deadlock two ranks sharing one
        # GPU (both stuck inside all_reduce), so the stream is drained first.  RCCL orders itself on the stream.
        self._drain = self.enabled and dist.get_backend(process_group) == "gloo"

    def _ready(self, t: torch.Tensor) -> None:
        if self._drain and t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def sum_moments(self, moments: torch.Tensor) -> int:
        """In-place SUM over ranks; returns the global transition count."""
        if self.active:
            self._ready(moments)
            self.dist.all_reduce(moments, op=self.dist.ReduceOp.SUM, group=self.group)
        return int(round(float(moments[0].item())))

    def sum_counts(self, values, device) -> list:
        """SUM over ranks of a few host integers (transition count, error flag ...): every rank gets the same totals, so
        rank-local decisions that would skip a later collective (an empty shard, a full buffer) are taken by ALL ranks
        together or by none."""
        vals = [int(v) for v in values]
        if not self.active:
            return vals
        t = torch.tensor(vals, dtype=torch.int64, device=device)
        self._ready(t)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.tolist()]

    def sum_grads(self, flat: torch.Tensor) -> None:
        if self.active:
            self._ready(flat)
            self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)

    def broadcast_params(self, flat: torch.Tensor, src: int = 0) -> None:
        """Replicas start from rank src's parameters (seeded init is already identical; this is a safety net)."""
        if self.active:
            self._ready(flat)
            self.dist.broadcast(flat, src=src, group=self.group)


def shard_range(n_total: int, rank: int, world_size: int):
    """Contiguous environment range [lo, hi) of a rank; the first ``n_total % world_size`` ranks get one more."""
    base, extra = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
