"""Batched evaluation matches between two networks.

The reference's only "play with the search" code is the UI's AI move (ui/app.py:108-121):
``prior = mcts.search(model, state, player, history, num_simulation, C_PUCT, use_Dirichlet)`` followed by
``prior.argmax(dim=1)``.  ``play_match`` plays ``n_games`` such games at once on one GPU, network A against
network B: every move of a game is that AI move made by the network whose turn it is (its own network
evaluates every leaf of its own search, as in the UI where one model owns one search), Dirichlet noise off
by default, ties to the lowest action index as ``argmax`` does on the reference's CPU tensors.
Games alternate colours: A moves first in even games, B in odd ones.
"""
from __future__ import annotations

from typing import Optional

import torch

from .arena import C_PUCT, EPSILON, TRAIN_NUM_SIMULATION, engine_constants, resolve_game
from .engine import Engine, EngineError


@torch.no_grad()
def play_match(game_cls, model_a: torch.nn.Module, model_b: torch.nn.Module, n_games: int,
               num_simulation: int = TRAIN_NUM_SIMULATION, c_puct: float = C_PUCT,
               device: Optional[torch.device] = None, record_moves: bool = False, fused: Optional[bool] = False) -> dict:
    """Returns {"a_wins", "b_wins", "draws", "results" (n_games,) from A's point of view in {+1, 0, -1},
    "plies" (n_games,), and with record_moves "moves": a list of (n_games,) int32 tensors, -1 once a game is over}.
    ``fused``: False = each network's own PyTorch forward on the rows it owns; True / None = the hand-written f32 network
    kernels (alphazero_amd.fused) reading the packed leaf batch, each network evaluating the rows flagged for it (None: only
    when kernels cover BOTH networks; True: required)."""
    if not torch.cuda.is_available():
        raise EngineError("play_match needs a GPU (no CPU fallback)")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    game_id, board = resolve_game(game_cls)
    eng = Engine(game_id, n_games, num_simulation, board_size=board, c_puct=c_puct, epsilon=EPSILON, device=dev,
                 **engine_constants(game_cls))
    d = eng.dims
    A = d.n_action
    models = (model_a.to(dev).eval(), model_b.to(dev).eval())
    evs = None
    if fused is None or fused:
        from .fused import FusedTowerEvaluator, SmallNetEvaluator, evaluator_available

        ok = all(evaluator_available(game_id, m, torch.float32) for m in models)
        if fused and not ok:
            raise EngineError("play_match(fused=True): no hand-written f32 kernel covers both networks")
        if ok:
            evs = tuple(FusedTowerEvaluator(m, "f32") if game_id == 2 else SmallNetEvaluator(m) for m in models)
            ring, leaf, _, _, _ = eng.leaf_buffers()
            flags = eng.leaf_flags()
            own_logits = (eng.new(n_games, A), eng.new(n_games, A))
    nn = eng.new_nn_input() if evs is None else None
    logits, values = eng.new(n_games, A), eng.new(n_games)
    actions = eng.new(n_games, dtype=torch.int32)
    done = torch.zeros(n_games, dtype=torch.uint8, device=dev)
    ones = torch.ones(n_games, device=dev)
    expo = torch.ones(n_games, A, device=dev)  # unused by a forced move, but the entry point wants the table
    a_first = (torch.arange(n_games, device=dev) % 2 == 0)  # A is PLAYER1 (+1, moves first) in even games
    results = torch.zeros(n_games, device=dev)
    ar = torch.arange(A, device=dev)
    moves = []
    eng.reset_games(None)
    while True:
        plies, player, active, _ = eng.game_info()
        act = active != 0
        if not bool(act.any()):
            break
        a_to_move = (player == 1) == a_first  # whose network owns this move's search
        idx = (torch.nonzero(act & a_to_move).reshape(-1), torch.nonzero(act & ~a_to_move).reshape(-1))
        eng.begin_search()
        for _ in range(num_simulation):
            eng.select_gather(nn)
            if evs is not None:
                # each network evaluates the rows that are its own AND need the network (its flag bytes); rows of the other
                # network keep stale data in this evaluator's outputs and are taken from the other's
                vs = []
                for k, ev in enumerate(evs):
                    fl = flags & (a_to_move if k == 0 else ~a_to_move).to(torch.uint8)
                    _, v = ev(None, logits_out=own_logits[k], packed=(ring, leaf, None, None, fl.data_ptr()), n_boards=n_games)
                    vs.append(v.view(-1))
                torch.where(a_to_move.view(-1, 1), own_logits[0], own_logits[1], out=logits)
                torch.where(a_to_move, vs[0], vs[1], out=values)
                eng.expand_backup(logits, values, None)
                continue
            for m, ix in zip(models, idx):
                if ix.numel():
                    p, v = m(nn.index_select(0, ix).to(next(m.parameters()).dtype))
                    logits.index_copy_(0, ix, p.float().reshape(ix.numel(), A))
                    values.index_copy_(0, ix, v.float().reshape(-1))
            eng.expand_backup(logits, values, None)
        visits = eng.new(n_games, A, dtype=torch.int32)
        eng.root_policy(visits=visits)
        # prior.argmax(dim=1): the most visited action, the lowest index among equals
        top = visits.max(1, keepdim=True).values
        forced = torch.where(visits == top, ar, A).min(1).values.to(torch.int32)
        forced = torch.where(act, forced, torch.full_like(forced, -1))
        eng.play_moves(ones, expo, forced, actions, done)
        if record_moves:
            moves.append(actions.clone())
        _, _, _, res = eng.game_info()
        fin = done != 0
        # Engine result = the LAST MOVER's point of view (training/self_play.py:86-89); the last mover is a_to_move
        results = torch.where(fin, torch.where(a_to_move, res, -res), results)
    eng.check()
    plies, _, _, _ = eng.game_info()
    out = {"a_wins": int((results > 0).sum()), "b_wins": int((results < 0).sum()), "draws": int((results == 0).sum()),
           "results": results.cpu(), "plies": plies.cpu()}
    if record_moves:
        out["moves"] = [m.cpu() for m in moves]
    return out
