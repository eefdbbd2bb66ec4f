"""Shared test helpers."""
import torch

M32 = 0xFFFFFFFF


def _mix32(x):
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & M32
    x = x ^ (x >> 16)
    return x


class HashNet(torch.nn.Module):
    """The synthetic 'hash network' (oracle/az_oracle.c az_oracle_eval_hash, csrc k_eval_hash) as a torch
    module obeying the model contract; integer arithmetic, so its f32 outputs are identical on CPU and GPU."""

    def __init__(self, n_action: int, scale: float = 5.0):
        super().__init__()
        self.n_action, self.scale = n_action, scale
        self.anchor = torch.nn.Parameter(torch.zeros(1))  # gives the module a device/dtype

    def forward(self, x):
        B = x.shape[0]
        dev = x.device
        code = x.reshape(B, -1).to(torch.float32).to(torch.int64) + 1
        i = torch.arange(code.shape[1], dtype=torch.int64, device=dev)
        h = _mix32((i * 2654435761 + code * 40503 + 12345) & M32).sum(1) & M32
        a = torch.arange(self.n_action, dtype=torch.int64, device=dev)
        r = _mix32((h[:, None] + a[None, :] * 0x9E3779B9 + 1) & M32)
        logits = self.scale * (((r >> 8) & 0xFFFF).to(torch.float32) * (1.0 / 65536.0) - 0.5)
        rv = _mix32(h ^ 0x00ABCDEF)
        value = ((rv >> 8) & 0xFFFF).to(torch.float32) * (1.0 / 32768.0) - 1.0
        return logits, value.view(B, 1)


class ListPool:
    """records put_tupule_experience calls (the pool boundary of training/self_play.py:122,132)"""

    def __init__(self):
        self.items = []

    def put_tupule_experience(self, exp):
        self.items.append(exp)
