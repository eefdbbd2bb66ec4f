"""Policy/value networks obeying the reference's model contract (nn_models/base.py:12-29):

    forward(x: (B, C_in, H, W) f32) -> (logits (B, A), value (B, 1) in [-1, 1])
           = policy_head(shared_layers(x)), value_head(shared_layers(x))

The engine only needs the contract: any reference ``BaseModel`` subclass works unchanged through its own
PyTorch-ROCm forward; the GomokuNet and QuickNet families (64 or 128 channels) and the two small nets are additionally
evaluated by the hand-written fused MFMA kernels (alphazero_amd.fused / csrc/az_tower*.hip, csrc/az_smallnet.hip).  The nets here exist because the
reference does not travel to the GPU box: they rebuild the reference's example architectures
(examples/tictactoe/model.py, examples/dots_and_boxes/model.py, examples/gomoku/model.py) with the
SAME parameter names, so a reference checkpoint (result/models/*/last_model.pth) loads with
``load_state_dict``; GomokuNet additionally takes the board size / width / depth as arguments
(the reference hard-codes 15 / 64 / 2, examples/gomoku/model.py:48-51), which is how BASELINE
configs 3 and 5 (9x9 board, deep tower) are expressed.
"""
from __future__ import annotations

import torch
from torch import nn

HISTORY_LEN = 8  # config/basic.py:10 (the reference's models read it when they are built: the history_len arguments below)


class DualHeadNet(nn.Module):
    """shared tower + two heads; attribute names follow the reference so checkpoints interchange."""

    _shared_layers: nn.Sequential
    _policy_head: nn.Sequential
    _value_head: nn.Sequential

    @property
    def shared_layers(self) -> nn.Sequential:
        return self._shared_layers

    def policy_head(self, x: torch.Tensor) -> torch.Tensor:
        return self._policy_head(x)

    def value_head(self, x: torch.Tensor) -> torch.Tensor:
        return self._value_head(x)

    def forward(self, x: torch.Tensor):
        h = self.shared_layers(x)
        return self.policy_head(h), self.value_head(h)


def _mlp_value(n_in: int) -> list[nn.Module]:
    return [nn.Linear(n_in, 64), nn.ReLU(), nn.Linear(64, 1), nn.Tanh()]


class TicTacToeNet(DualHeadNet):
    """(B,9,3,3) -> (B,9),(B,1); 154,186 parameters."""

    def __init__(self, history_len: int = HISTORY_LEN) -> None:
        super().__init__()
        c = 64
        self._shared_layers = nn.Sequential(nn.Conv2d(history_len + 1, c, 3, padding=1), nn.ReLU(),
                                            nn.Conv2d(c, c, 3, padding=1), nn.ReLU())
        self._policy_head = nn.Sequential(nn.Flatten(1), nn.Linear(c * 9, 128), nn.ReLU(), nn.Linear(128, 9))
        self._value_head = nn.Sequential(nn.Flatten(1), *_mlp_value(c * 9))


class DotsAndBoxesNet(DualHeadNet):
    """(B,25,4,4) -> (B,24),(B,1); 53,532 parameters.  box_rows / box_cols: the reference's model reads its plane size and action
    count from the game class's constants (examples/dots_and_boxes/model.py:27-29, :44) -- the same network for another board."""

    def __init__(self, box_rows: int = 3, box_cols: int = 0, history_len: int = HISTORY_LEN) -> None:
        super().__init__()
        r = int(box_rows)
        cc = int(box_cols) or r
        c, hw, n_action = 64, (r + 1) * (cc + 1), (r + 1) * cc + r * (cc + 1)
        self._shared_layers = nn.Sequential(nn.Conv2d(history_len * 3 + 1, c, 3, padding=1), nn.ReLU(),
                                            nn.Conv2d(c, c, 3, padding=1), nn.ReLU())
        self._policy_head = nn.Sequential(nn.Conv2d(c, 2, 1), nn.ReLU(), nn.Flatten(1), nn.Linear(2 * hw, n_action))
        self._value_head = nn.Sequential(nn.Conv2d(c, 1, 1), nn.ReLU(), nn.Flatten(1), *_mlp_value(hw))


class _Res(nn.Module):
    def __init__(self, c: int) -> None:
        super().__init__()
        self._conv1 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self._bn1 = nn.BatchNorm2d(c)
        self._conv2 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self._bn2 = nn.BatchNorm2d(c)
        self._relu = nn.ReLU()

    def forward(self, x):
        y = self._relu(self._bn1(self._conv1(x)))
        y = self._bn2(self._conv2(y))
        return self._relu(y + x)


class GomokuNet(DualHeadNet):
    """ResNet tower; defaults = the reference GomokuModel (15x15, 64 ch, 2 blocks, 233,257 params)."""

    def __init__(self, board_size: int = 15, channels: int = 64, blocks: int = 2, history_len: int = HISTORY_LEN) -> None:
        super().__init__()
        a = board_size * board_size
        self.board_size = board_size
        self._shared_layers = nn.Sequential(
            nn.Conv2d(history_len + 1, channels, 3, padding=1, bias=False), nn.BatchNorm2d(channels), nn.ReLU(),
            *[_Res(channels) for _ in range(blocks)])
        self._policy_head = nn.Sequential(nn.Conv2d(channels, 1, 1, bias=False), nn.BatchNorm2d(1), nn.ReLU(),
                                          nn.Flatten(1), nn.Linear(a, a))
        self._value_head = nn.Sequential(nn.Conv2d(channels, 2, 1, bias=False), nn.BatchNorm2d(2), nn.ReLU(),
                                         nn.Flatten(1), *_mlp_value(2 * a))

    def flops_per_eval(self) -> float:
        """multiply-adds x2 of one forward (convs + linears), for MFMA utilisation reporting"""
        total = 0.0
        hw = self.board_size ** 2
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                total += 2.0 * hw * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1]
            elif isinstance(m, nn.Linear):
                total += 2.0 * m.in_features * m.out_features
        return total


class _QuickResBlock(nn.Module):
    """nn_models/resnet_blocks.py:28-40: ``x + Sequential((conv, BN, ReLU) x layers)`` -- no activation after the sum."""

    def __init__(self, channels: int, layers: int) -> None:
        super().__init__()
        mods: list[nn.Module] = []
        for _ in range(layers):
            mods += [nn.Conv2d(channels, channels, 3, padding=1), nn.BatchNorm2d(channels), nn.ReLU()]
        self.model = nn.Sequential(*mods)

    def forward(self, x):
        return x + self.model(x)


class QuickNet(DualHeadNet):
    """The reference's QuickModel (nn_models/quick_model.py:29-240, built from config/quick_model_config.py) for ANY board:
    the reference hard-wires it to tic-tac-toe through GAME_STATE_DIM / NUM_ACTION (config/quick_model_config.py:7-10), so
    "QuickModel scaled up" (BASELINE config 5) has no instance there.  Same layer sequence and parameter names as the
    reference's defaults: transition conv3x3(C_in -> channels)+BN+ReLU, ``blocks`` residual blocks of ``layers`` x
    (conv3x3, BN, ReLU) with the sum taken AFTER the last ReLU; policy head conv3x3(channels -> 2)+BN+ReLU, Linear(2HW -> 64),
    ReLU, Linear(64 -> A); value head conv3x3(channels -> 1)+BN+ReLU, Linear(HW -> 64), ReLU, Linear(64 -> 1), tanh applied
    in ``value_head`` (:131-133).  Evaluated by its own PyTorch-ROCm forward or, at 64 / 128 channels on the 9x9 / 15x15 boards,
    by the QuickModel instantiations of the fused towers (fused.FusedTowerEvaluator: f32 = the reference's precision, bf16)."""

    def __init__(self, board_size: int = 15, in_channels: int = HISTORY_LEN + 1, channels: int = 128, blocks: int = 1,
                 layers: int = 5, n_action: int | None = None, features: int = 64) -> None:
        super().__init__()
        hw = board_size * board_size
        a = n_action if n_action is not None else hw
        self.board_size = board_size
        self._shared_layers = nn.Sequential(nn.Conv2d(in_channels, channels, 3, padding=1), nn.BatchNorm2d(channels), nn.ReLU(),
                                            *[_QuickResBlock(channels, layers) for _ in range(blocks)])
        self._policy_head = nn.Sequential(nn.Conv2d(channels, 2, 3, padding=1), nn.BatchNorm2d(2), nn.ReLU(), nn.Flatten(1),
                                          nn.Linear(2 * hw, features), nn.ReLU(), nn.Linear(features, a))
        self._value_head = nn.Sequential(nn.Conv2d(channels, 1, 3, padding=1), nn.BatchNorm2d(1), nn.ReLU(), nn.Flatten(1),
                                         nn.Linear(hw, features), nn.ReLU(), nn.Linear(features, 1))

    def value_head(self, x: torch.Tensor) -> torch.Tensor:
        return torch.tanh(self._value_head(x))

    flops_per_eval = GomokuNet.flops_per_eval


def flat_parameters(model: nn.Module) -> list[torch.Tensor]:
    """parameters + buffers in state_dict order: what a weight publish has to move
    (training/alphazero_trainer.py:53-66 moves the whole state_dict)."""
    return [t for t in model.state_dict().values() if t.is_floating_point() or t.dtype == torch.long]
