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
    def publish(self, force: bool = False) -> int:
        """-> bytes broadcast (0 when there is nobody to publish to; force=True runs the collective even with one rank)"""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
            return 0
        off = 0
        if dist.get_rank() == self.src:
            for t in self.tensors:
                self.flat[off:off + t.numel()].copy_(t.reshape(-1))
                off += t.numel()
        if dist.get_backend() == "gloo" and self.flat.is_cuda:  # rehearsal backend: the collective runs on a host copy
            h = self.flat.cpu()
            dist.broadcast(h, src=self.src)
            self.flat.copy_(h)
        else:
            dist.broadcast(self.flat, src=self.src)  # RCCL over xGMI
        if dist.get_rank() != self.src:
            off = 0
            for t in self.tensors:
                t.copy_(self.flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        return self.flat.numel() * 4


def games_for_rank(n_games: int, rank: int, world: int) -> range:
    """Game ids owned by a rank: game g lives on rank g % world; no game ever moves."""
    return range(rank, n_games, world)


class DeviceSampleRing:
    """The reference's ring buffer (utils/share_ring_buffer.py:39-46,129-160) resident in HBM, for a learner on
    the same GPU: states (cap,1,C,H,W), prior (cap,1,A), values (cap,1), overwrite-oldest,
    ``sample`` draws with replacement from [0, size).  ``put_games`` runs the 8-fold augmentation on the device
    (az_augment_samples) straight into ring layout, so no sample ever visits the host."""

    def __init__(self, engine, capacity: int):
        d = engine.dims
        if capacity < 8:
            raise ValueError("capacity must be >= 8")
        self.e, self.capacity = engine, capacity
        dev = engine.device
        self._states = torch.zeros((capacity, 1, d.nn_channels, d.height, d.width), device=dev)
        self._prior = torch.zeros((capacity, 1, d.n_action), device=dev)
        self._values = torch.zeros((capacity, 1), device=dev)
        self.write_idx = 0
        self._size = 0
        self.total_written = 0

    def _advance(self, rows: int) -> None:
        self.write_idx = (self.write_idx + rows) % self.capacity
        self._size = min(self._size + rows, self.capacity)
        self.total_written += rows

    def put_games(self, states, pis, zs, augment: bool = True) -> None:
        """device tensors from Engine.emit_samples: states (n, C*H*W), pis (n, A), zs (n,)"""
        if not augment:
            return self.put_batch(states, pis, zs)
        n = int(states.shape[0])
        step = max(self.capacity // 8, 1)  # never more than one lap per launch
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            self.e.augment_samples(states[lo:hi].contiguous(), pis[lo:hi].contiguous(), zs[lo:hi].contiguous(),
                                   out=(self._states, self._prior, self._values), ring_capacity=self.capacity,
                                   ring_write_idx=self.write_idx)
            self._advance(8 * (hi - lo))

    def put_batch(self, states, pis, zs) -> None:
        n = int(states.shape[0])
        done = 0
        while done < n:
            k = min(n - done, self.capacity - self.write_idx)
            sl = slice(self.write_idx, self.write_idx + k)
            self._states[sl].view(k, -1).copy_(states[done:done + k].reshape(k, -1))
            self._prior[sl].view(k, -1).copy_(pis[done:done + k])
            self._values[sl].view(k).copy_(zs[done:done + k])
            self._advance(k)
            done += k

    def sample(self, batch_size: int, timeout=None):
        if self._size == 0:
            return None
        k = min(batch_size, self._size)
        idxs = torch.randint(0, self._size, (k,), dtype=torch.int64, device=self._states.device)
        return self._states[idxs].squeeze(1), self._prior[idxs].squeeze(1), self._values[idxs]

    def size(self) -> int:
        return self._size

    def __len__(self) -> int:
        return self.capacity


class SharedRingWriter:
    """Bulk hand-off of finished games to the reference's ``SharedRingBufferExperiencePool`` used UNCHANGED
    (utils/share_ring_buffer.py).  The reference reaches that pool one position at a time --
    ``put_tupule_experience`` (:119-127) -> ``get_enhanced_data`` -> eight ``put`` calls (:56-88), each taking the lock
    and copying one row -- which at thousands of finished games per second is the bottleneck of the whole drop-in.
    This writer does what those calls would have done, for a whole batch at once: the 8-fold augmentation on the GPU
    (az_augment_samples, the reference's order), one device-to-host copy, then -- under the pool's own lock -- the rows
    go into ``_states / _prior / _values`` at ``_write_idx`` with wrap-around, and ``_write_idx / _read_idx / _size /
    _total_written`` advance exactly as 8 n sequential ``put`` calls would have left them; waiting samplers are woken.

    ``writer = SharedRingWriter(pool, engine); writer.put_batch(states, pis, zs)`` with the tensors of
    ``Engine.emit_samples`` (device or host; put order).  ``ChessArena.self_play_many`` builds one by itself when it is
    handed such a pool."""

    # which of the 8 rows az_augment_samples writes per sample (rot90 k, then its flip, k = 0..3) the reference's enhancer
    # returns under its config switches (utils/data_enhancer.py:29-45): rotation + flip: all; rotation only: the four rotations;
    # flip only: the flipped sample ALONE (the reference does not keep the original there); enhancement off: the sample itself
    ROWS = {"rot+flip": (0, 1, 2, 3, 4, 5, 6, 7), "rot": (0, 2, 4, 6), "flip": (1,), "none": (0,)}

    @staticmethod
    def mode_of(pool) -> str:
        """the augmentation the pool's own ``put_tupule_experience`` would apply: the config switches its module imported
        (utils/share_ring_buffer.py:119-127 reads USE_DATA_ENHANCEMENT; the enhancer USE_ROATATION / USE_FLIP,
        config/utils_config.py:27-55); 8-fold when the module does not say"""
        import sys

        mod = sys.modules.get(type(pool).__module__)
        if not bool(getattr(mod, "USE_DATA_ENHANCEMENT", True)):
            return "none"
        rot, flip = bool(getattr(mod, "USE_ROATATION", True)), bool(getattr(mod, "USE_FLIP", True))
        return "rot+flip" if rot and flip else "rot" if rot else "flip" if flip else "none"

    def __init__(self, pool, engine, augment=None):
        """augment: None = as the pool's own config says (mode_of), True = 8-fold, False = none, or one of ROWS' keys"""
        for name in ("_states", "_prior", "_values", "_write_idx", "_read_idx", "_size", "_total_written", "_lock", "_not_empty"):
            if not hasattr(pool, name):
                raise TypeError(f"not a SharedRingBufferExperiencePool-like object: no attribute {name}")
        self.mode = self.mode_of(pool) if augment is None else ("rot+flip" if augment is True else "none" if augment is False else str(augment))
        if self.mode not in self.ROWS:
            raise ValueError(f"augment: one of {sorted(self.ROWS)}")
        self.pool, self.e, self.augment = pool, engine, self.mode != "none"
        self.capacity = int(pool._states.shape[0])
        d = engine.dims
        if pool._states[0].numel() != d.nn_numel or pool._prior[0].numel() != d.n_action:
            raise ValueError("the pool's row shapes do not match the engine's game")

    @staticmethod
    def accepts(pool) -> bool:
        return all(hasattr(pool, n) for n in ("_states", "_prior", "_values", "_write_idx", "_read_idx", "_size",
                                               "_total_written", "_lock", "_not_empty"))

    def put_batch(self, states, pis, zs) -> int:
        """states / pis / zs: device tensors (no copy; what ChessArena.self_play_many hands over) or host tensors.
        -> rows written (8 per position with the full augmentation)"""
        dev = self.e.device
        n = int(states.shape[0])
        if n == 0:
            return 0
        states = states.reshape(n, -1).to(dev, torch.float32).contiguous()
        pis = pis.reshape(n, -1).to(dev, torch.float32).contiguous()
        zs = zs.reshape(n).to(dev, torch.float32).contiguous()
        if self.augment:
            states, pis, zs = self.e.augment_samples(states, pis, zs)
            if self.mode != "rot+flip":  # the subset of the 8 the reference's switches leave, in its order
                sel = (8 * torch.arange(n, device=dev).view(-1, 1) + torch.tensor(self.ROWS[self.mode], device=dev).view(1, -1)).reshape(-1)
                states, pis, zs = states[sel], pis[sel], zs[sel]
        rows = int(states.shape[0])
        s_h, p_h, z_h = states.cpu(), pis.cpu(), zs.cpu()
        p, cap = self.pool, self.capacity
        with p._lock:
            w, size = int(p._write_idx.value), int(p._size.value)
            done = 0
            while done < rows:  # in put order, so a batch longer than the ring leaves its newest rows, as puts would
                k = min(rows - done, cap - w)
                p._states[w:w + k].view(k, -1).copy_(s_h[done:done + k])
                p._prior[w:w + k].view(k, -1).copy_(p_h[done:done + k])
                p._values[w:w + k].view(k).copy_(z_h[done:done + k])
                w = (w + k) % cap
                done += k
            overwritten = max(0, size + rows - cap)  # puts that found the ring full: each moved the read index on
            p._read_idx.value = (int(p._read_idx.value) + overwritten) % cap
            p._write_idx.value = w
            p._size.value = min(size + rows, cap)
            p._total_written.value = int(p._total_written.value) + rows
            p._not_empty.notify_all()
        return rows
