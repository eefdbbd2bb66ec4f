"""``MCTS(game_cls).search(...)``: the reference's single-position search entry point
(core/MCTS_alphazero.py:112-178), served by the HIP engine with one slot.

Same signature, same return value -- a (1, A) float32 tensor ``visits / visits.sum()`` -- and the same
consumption of torch's global CPU generator (one Dirichlet draw per search when ``use_Dirichlet``), so
callers such as ``ui/app.py:110`` or ``training/self_play.py:46`` can switch by changing an import.
This is the batch-1 latency path; throughput comes from ``ChessArena.self_play_many``.
"""
from __future__ import annotations

from typing import Optional

import torch

from .arena import ALPHA, C_PUCT, EPSILON, TRAIN_NUM_SIMULATION, engine_constants, resolve_game, resolve_win_length
from .engine import Engine, EngineError


class MCTS:
    def __init__(self, game_cls, device: Optional[torch.device] = None) -> None:
        self.game_cls = game_cls
        self.game_id, self.board_size = resolve_game(game_cls)
        self.win_length = resolve_win_length(game_cls)
        self._constants = engine_constants(game_cls)
        self.device = torch.device(device) if device is not None else None
        self._engines: dict = {}

    def _engine(self, num_simulation: int, c_puct: float, device: torch.device) -> Engine:
        key = (num_simulation, float(c_puct), str(device))
        if key not in self._engines:
            self._engines[key] = Engine(self.game_id, 1, num_simulation, board_size=self.board_size, c_puct=c_puct,
                                        epsilon=EPSILON, device=device, **self._constants)
        return self._engines[key]

    def search(self, model: torch.nn.Module, root_state: torch.Tensor, root_player: int,
               history_state: Optional[torch.Tensor] = None, num_simulation: int = TRAIN_NUM_SIMULATION,
               c_puct: float = C_PUCT, use_Dirichlet: bool = True) -> torch.Tensor:
        if history_state is None:
            # USE_HISTORY is True in the reference (config/basic.py:9); same error as :138-139
            raise ValueError("history_state is required (USE_HISTORY)")
        p = next(model.parameters(), None)
        model_device = p.device if p is not None else torch.device("cpu")
        if self.device is not None:
            dev = self.device
        elif model_device.type == "cuda":
            dev = model_device
        elif torch.cuda.is_available():
            dev = torch.device("cuda", torch.cuda.current_device())
        else:
            raise EngineError("MCTS.search needs a GPU (no CPU fallback)")
        eng = self._engine(num_simulation, c_puct, dev)
        d = eng.dims
        hist = history_state.detach().to("cpu", torch.float32).reshape(eng.history_len, -1)
        if not torch.equal(hist[-1], root_state.detach().to("cpu", torch.float32).reshape(-1)):
            raise ValueError("the newest history entry must be root_state")
        eng.set_position(0, hist, int(root_player))
        nn = eng.new_nn_input()
        logits, values = eng.new(1, d.n_action), eng.new(1)
        noise = None
        if use_Dirichlet:  # Dirichlet(torch.ones_like(policy) * ALPHA).sample(), core/MCTS_alphazero.py:187
            from torch.distributions import Dirichlet

            noise = Dirichlet(torch.ones(1, d.n_action) * ALPHA).sample().to(dev)
        eng.begin_search()
        for _ in range(num_simulation):
            eng.select_gather(nn)
            with torch.inference_mode():
                pol, val = model(nn.to(model_device))
            logits.copy_(pol.reshape(1, -1))
            values.copy_(val.reshape(-1))
            eng.expand_backup(logits, values, noise)
        pi = eng.root_policy()
        eng.check()
        return pi.to(model_device if model_device.type == "cuda" else "cpu")
