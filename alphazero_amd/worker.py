"""A self-play worker that owns a GPU, with the signature of the reference's ``_self_player_worker``
(training/alphazero_trainer.py:164-221), so ``_multi_process_parallel`` (:395-456) can spawn it in place of the
CPU workers: same seeding, same shared-``state_dict``-under-a-lock weight refresh
(TrainerUtils.sync_model_from_shared, :53-57), same temperature schedule -- but the games of a worker run
``n_slots`` at a time on its GPU instead of one by one."""
from __future__ import annotations

import logging
import random
from typing import Any

import torch

from .arena import ChessArena, reference_tao

logger = logging.getLogger(__name__)

SELF_PLAY_UPDATE_MODEL_FREQUENCY = 10  # config/train_config.py:27 (games per worker between weight refreshes)


def sync_model_from_shared(model: torch.nn.Module, model_state: dict, lock: Any) -> None:
    """training/alphazero_trainer.py:53-57: clone the shared tensors under the lock, then load them"""
    with lock:
        state = {k: v.clone() for k, v in model_state.items()}
    model.load_state_dict(state)


def gpu_self_player_worker(model_cls, game_cls, model_state, experience_pool, num_self_play_games: int, worker_id: int,
                           model_lock: Any, *, seed_bias: int = 0, n_slots: int = 1024, num_simulation: int = 80,
                           device: torch.device | None = None, trace: list | None = None, fused: bool | None = False,
                           use_graph: bool = False, nn_dtype: torch.dtype = torch.float32) -> None:
    """``fused`` / ``use_graph`` / ``nn_dtype``: as for ChessArena (fused=None: the hand-written f32 / bf16 network kernels when
    they cover ``model_cls``, else its own forward)."""
    try:
        seed = seed_bias + worker_id
        random.seed(seed)
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        dev = device or torch.device("cuda", worker_id % max(torch.cuda.device_count(), 1))
        # the worker's seed also keys its games' counter-based random rows (arena.seed): workers get distinct streams as in the
        # reference, and a worker's game k plays the same moves whatever n_slots is
        arena = ChessArena(model_cls, game_cls, dev, n_slots=n_slots, num_simulation=num_simulation, fused=fused, use_graph=use_graph,
                           nn_dtype=nn_dtype, seed=seed)
        sync_model_from_shared(arena.model, model_state, model_lock)
        # one "generation" = n_slots concurrent games at the temperature the reference's worker would use for its
        # generation-th game; weights are refreshed every SELF_PLAY_UPDATE_MODEL_FREQUENCY generations' worth of games
        chunk = max(n_slots, 1) * SELF_PLAY_UPDATE_MODEL_FREQUENCY
        played = 0
        while played < num_self_play_games:
            n = min(chunk, num_self_play_games - played)
            arena.self_play_many(n, lambda i: reference_tao(i // max(n_slots, 1)), experience_pool, trace=trace)
            played += n
            if played < num_self_play_games:
                sync_model_from_shared(arena.model, model_state, model_lock)
        logger.info("self-play worker %d finished %d games", worker_id, played)
    except Exception as exc:  # the reference logs and lets the process end (:220-221)
        logger.error("self-play worker %d failed: %s", worker_id, exc, exc_info=True)
        raise
