"""Host-side landing zone for finished games' samples, and the RCCL weight publish.

``HostSampleRing`` is a pinned-memory ring with the layout of the reference's shared ring buffer
(utils/share_ring_buffer.py:39-46: state (cap,1,C,H,W), prior (cap,1,A), value (cap,1)); it only
receives batches (``put_batch``) by one asynchronous D2H copy per finished batch of games -- sampling,
augmentation and locking stay in the reference's unchanged pool classes.
"""
from __future__ import annotations

import torch


class HostSampleRing:
    def __init__(self, capacity: int, nn_shape: tuple, n_action: int, pin: bool = True):
        self.capacity = capacity
        self.states = torch.empty((capacity, 1, *nn_shape), dtype=torch.float32, pin_memory=pin)
        self.priors = torch.empty((capacity, 1, n_action), dtype=torch.float32, pin_memory=pin)
        self.values = torch.empty((capacity, 1), dtype=torch.float32, pin_memory=pin)
        self.write_idx = 0
        self.total = 0

    def put_batch(self, states, pis, zs) -> None:
        """states (n, C*H*W) or (n,C,H,W), pis (n,A), zs (n,): device or host tensors, put order kept."""
        n = int(states.shape[0])
        done = 0
        while done < n:
            k = min(n - done, self.capacity - self.write_idx)
            sl = slice(self.write_idx, self.write_idx + k)
            self.states[sl].view(k, -1).copy_(states[done:done + k].reshape(k, -1), non_blocking=True)
            self.priors[sl].view(k, -1).copy_(pis[done:done + k], non_blocking=True)
            self.values[sl].view(k).copy_(zs[done:done + k], non_blocking=True)
            self.write_idx = (self.write_idx + k) % self.capacity
            done += k
        self.total += n

    def size(self) -> int:
        return min(self.total, self.capacity)


class WeightPublisher:
    """Learner -> self-play ranks weight push: the reference clones a shared-memory state_dict under a
    lock (training/alphazero_trainer.py:53-66); here it is ONE flat buffer and ONE RCCL broadcast over
    xGMI (sub-MB for the reference nets => latency bound, so a single collective, no bucketing)."""

    def __init__(self, model: torch.nn.Module, src: int = 0):
        self.model, self.src = model, src
        self.tensors = [t for t in model.state_dict().values() if t.is_floating_point()]
        n = sum(t.numel() for t in self.tensors)
        dev = self.tensors[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def publish(self) -> int:
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0
        off = 0
        if dist.get_rank() == self.src:
            for t in self.tensors:
                self.flat[off:off + t.numel()].copy_(t.reshape(-1))
                off += t.numel()
        dist.broadcast(self.flat, src=self.src)
        if dist.get_rank() != self.src:
            off = 0
            for t in self.tensors:
                t.copy_(self.flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        return self.flat.numel() * 4


def games_for_rank(n_games: int, rank: int, world: int) -> range:
    """Game ids owned by a rank: game g lives on rank g % world; no game ever moves."""
    return range(rank, n_games, world)
