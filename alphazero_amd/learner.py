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
