"""The SGD step of the reference (training/nn_trainer.py:49-94) on batches that already live on the GPU
(SURVEY.md 8f-3).  Stock PyTorch: loss = -(pi * log_softmax(p)).sum(1).mean() + mse(v.squeeze(), z.squeeze())."""
from __future__ import annotations

import torch


def calculate_loss(model: torch.nn.Module, states, target_policies, target_values):
    pred_policies, pred_values = model(states)
    log_policies = torch.log_softmax(pred_policies, dim=1)
    policy_loss = -torch.sum(target_policies * log_policies, dim=1).mean()
    value_loss = torch.nn.functional.mse_loss(pred_values.squeeze(), target_values.squeeze())
    return policy_loss + value_loss, policy_loss.detach(), value_loss.detach()


def train_step(model: torch.nn.Module, optim: torch.optim.Optimizer, pool, batch_size: int = 64):
    """one Trainer.train(): zero_grad, loss on pool.sample(batch_size), backward, step"""
    batch = pool.sample(batch_size, timeout=1.0)
    if batch is None:
        raise ValueError("the experience pool holds no data")
    dev = next(model.parameters()).device
    states, pis, zs = (t.to(dev) for t in batch)
    model.train()
    optim.zero_grad()
    loss, pl, vl = calculate_loss(model, states, pis, zs)
    loss.backward()
    optim.step()
    model.eval()
    return float(loss.detach()), float(pl), float(vl)


class DataParallelLearner:
    """``train_step`` on every rank's own batch with the gradients averaged across ranks (BASELINE
    config 5): one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

    Gradients live in a few large flat buckets (parameters' ``.grad`` are views into them), filled in backward order;
    a bucket's all-reduce is launched asynchronously from the hook of its last gradient, so it runs under the rest of
    backward.  xGMI is point to point (7 links per GPU), a ring all-reduce is per-link bound, so buckets are sized in
    tens of MB (few, large collectives): the reference nets (< 1 MB of gradients) go out as ONE collective.
    The reference has no counterpart: its learner is one process (training/nn_trainer.py:36-94)."""

    def __init__(self, model: torch.nn.Module, optim: torch.optim.Optimizer, bucket_bytes: int = 32 << 20,
                 process_group=None, force_collectives: bool = False):
        """force_collectives: run the hook / all-reduce path even with one rank (a process group must be initialised):
        how a one-GPU box exercises the code every rank of an N-GPU job runs."""
        import torch.distributed as dist

        self.model, self.optim, self.group = model, optim, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())
        params = [p for p in model.parameters() if p.requires_grad]
        self.buckets: list[torch.Tensor] = []
        self._bucket_of: dict[int, int] = {}
        self._need: list[int] = []
        cur: list[torch.nn.Parameter] = []
        cur_n = 0

        def close():
            nonlocal cur, cur_n
            if not cur:
                return
            flat = torch.zeros(cur_n, dtype=cur[0].dtype, device=cur[0].device)
            off = 0
            for p in cur:
                p.grad = flat[off:off + p.numel()].view_as(p)
                self._bucket_of[id(p)] = len(self.buckets)
                off += p.numel()
            self.buckets.append(flat)
            self._need.append(len(cur))
            cur, cur_n = [], 0

        for p in reversed(params):  # backward produces gradients roughly in reverse registration order
            if cur and (cur_n * p.element_size() >= bucket_bytes or p.dtype != cur[0].dtype):
                close()
            cur.append(p)
            cur_n += p.numel()
        close()
        self._left = list(self._need)
        self._handles: list = []
        self._sent: list[bool] = [False] * len(self.buckets)
        self.collectives_issued = 0
        if self.collective:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _send(self, b: int) -> None:
        import torch.distributed as dist

        self._sent[b] = True
        self.collectives_issued += 1
        self._handles.append(dist.all_reduce(self.buckets[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b = self._bucket_of[id(p)]
        self._left[b] -= 1
        if self._left[b] == 0:
            self._send(b)

    def step(self, pool, batch_size: int = 64):
        """one data-parallel Trainer.train(): every rank samples its own batch_size rows from its own pool"""
        batch = pool.sample(batch_size, timeout=1.0)
        if batch is None:
            raise ValueError("the experience pool holds no data")
        dev = next(self.model.parameters()).device
        states, pis, zs = (t.to(dev) for t in batch)
        self.model.train()
        for flat in self.buckets:  # zero_grad that keeps the views
            flat.zero_()
        self._left = list(self._need)
        self._handles = []
        self._sent = [False] * len(self.buckets)
        loss, pl, vl = calculate_loss(self.model, states, pis, zs)
        loss.backward()
        if self.collective:
            # A bucket whose hooks did not all fire (a parameter that took no part in this step's graph: its gradient
            # stays zero) is reduced here, in bucket order, so that every rank issues every bucket exactly once per
            # step.  (As with any data-parallel scheme the SET of unused parameters must be the same on every rank.)
            for b in range(len(self.buckets)):
                if not self._sent[b]:
                    self._send(b)
            for h in self._handles:
                h.wait()
            if self.world > 1:
                for flat in self.buckets:
                    flat.div_(self.world)
        self.optim.step()
        self.model.eval()
        return float(loss.detach()), float(pl), float(vl)
