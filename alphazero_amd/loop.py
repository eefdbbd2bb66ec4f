"""Self-play + training on the same GPUs (BASELINE config 5; SURVEY.md 8f-1..3 put together).

Every rank (one process per GPU) runs the same loop: one move of batched self-play for its own games ->
finished games are augmented on the device straight into an HBM ring (no sample visits the host) ->
``train_steps_per_move`` data-parallel SGD steps (gradients averaged over ranks, RCCL all-reduce) ->
the self-play network is refreshed from the learner's weights.  What the reference does with N CPU worker
processes, one trainer process, a shared-memory state_dict and a shared-memory ring
(training/alphazero_trainer.py:395-456) happens here inside each GPU process; the only inter-GPU traffic
is the gradient all-reduce and a periodic weight/buffer broadcast from rank 0 (BatchNorm statistics are
rank-local otherwise), the counterpart of ``sync_shared_from_model`` (:60-66, every 300 steps).
"""
from __future__ import annotations

import copy
from typing import Optional

import torch

from .arena import SelfPlayRunner, reference_tao
from .engine import Engine
from .learner import DataParallelLearner
from .samples import DeviceSampleRing, WeightPublisher

UPDATE_MODEL_FREQUENCY = 300  # config/train_config.py:28 (train steps between publishes)


class SelfPlayTrainLoop:
    def __init__(self, model: torch.nn.Module, game: int, board_size: int, n_slots: int, num_sims: int,
                 device: torch.device, *, ring_capacity: int = 1 << 18, batch_size: int = 512,
                 train_steps_per_move: int = 1, min_samples: int = 4096, lr: float = 1e-3,
                 nn_dtype: torch.dtype = torch.bfloat16, fused: Optional[bool] = None,
                 publish_every: int = UPDATE_MODEL_FREQUENCY, force_collectives: bool = False,
                 engine=None, runner=None, ring=None):
        """engine / runner / ring: prebuilt parts (the CPU tests of the multi-rank control flow hand in stand-ins that
        need no GPU); by default they are built here from the other arguments."""
        self.device = device
        self.force_collectives = force_collectives
        self.master = model.to(device).float()  # the learner's f32 weights
        self.optim = torch.optim.Adam(self.master.parameters(), lr=lr)
        self.learner = DataParallelLearner(self.master, self.optim, force_collectives=force_collectives)
        self.publisher = WeightPublisher(self.master, src=0)
        self.publisher.publish(force=force_collectives)  # every rank starts from rank 0's weights
        self.player = copy.deepcopy(self.master).to(nn_dtype).eval()  # the self-play copy (bf16 by default)
        self.engine = engine if engine is not None else \
            Engine(game, n_slots, num_sims, board_size=board_size, nn_dtype=nn_dtype, device=device)
        if runner is None:
            if fused is None:
                from .fused import FusedTowerEvaluator

                fused = FusedTowerEvaluator.precision_for(self.player, nn_dtype) is not None
            runner = SelfPlayRunner(self.engine, self.player, num_sims, fused=bool(fused))
        self.runner = runner
        self.ring = ring if ring is not None else DeviceSampleRing(self.engine, ring_capacity)
        self.batch_size, self.train_steps_per_move = batch_size, train_steps_per_move
        self.min_samples, self.publish_every = min_samples, publish_every
        self.n_slots = n_slots
        self.inv_tao_h = torch.full((n_slots,), 1.0 / reference_tao(0))
        self.inv_tao = self.inv_tao_h.to(device)
        self.games_started = n_slots
        self.games_finished = self.samples = self.train_steps = self.moves = 0
        self.last_loss = None
        self.loss_history: list[float] = []

    @torch.no_grad()
    def _refresh_player(self) -> None:
        dst = self.player.state_dict()
        for k, v in self.master.state_dict().items():
            dst[k].copy_(v)
        if getattr(self.runner, "fused", None) is not None:
            self.runner.fused.refresh()

    def _all_ranks_ready(self) -> bool:
        """Training is a collective (gradient all-reduce, weight broadcast), so the decision to train this move must be
        one too: every rank trains exactly when EVERY rank's ring holds min_samples.  Ranks play different games and
        cross the threshold on different moves; a rank-local test would leave the early ranks blocked in an all-reduce
        their peers never post.  One 4-byte MIN all-reduce per move (a move is hundreds of ms of search)."""
        import torch.distributed as dist

        ready = self.ring.size() >= self.min_samples
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collectives):
            dev = self.device if torch.device(self.device).type == "cuda" else torch.device("cpu")
            flag = torch.tensor([1 if ready else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ready = bool(int(flag.item()))
        return ready

    def move(self) -> None:
        """one self-play move of every game, then the learner's share of work"""
        eng = self.engine
        _, done = self.runner.move_round(self.inv_tao)
        done_h = done.cpu()
        idx = torch.nonzero(done_h).reshape(-1)
        if idx.numel():
            plies, _, _, _ = eng.game_info()
            states, pis, zs, _ = eng.emit_samples(idx.to(torch.int32), plies.cpu()[idx])
            self.ring.put_games(states, pis, zs)  # 8-fold augmentation on the device, ring layout
            self.samples += int(states.shape[0])
            self.games_finished += int(idx.numel())
            for s in idx.tolist():
                self.inv_tao_h[s] = 1.0 / reference_tao(self.games_started // self.n_slots)
                self.games_started += 1
            self.inv_tao.copy_(self.inv_tao_h, non_blocking=True)
            eng.reset_games(done)
        self.moves += 1
        if self._all_ranks_ready():  # the same answer on every rank => the same number of collectives below
            for _ in range(self.train_steps_per_move):
                self.last_loss = self.learner.step(self.ring, self.batch_size)
                self.loss_history.append(self.last_loss[0])
                self.train_steps += 1  # identical on every rank (see _all_ranks_ready), so is the publish schedule
                if self.publish_every and self.train_steps % self.publish_every == 0:
                    self.publisher.publish(force=self.force_collectives)
            self._refresh_player()

    def run(self, n_moves: int) -> dict:
        for _ in range(n_moves):
            self.move()
        self.runner.check() if hasattr(self.runner, "check") else self.engine.check()
        return {"moves": self.moves, "games": self.games_finished, "samples": self.samples,
                "train_steps": self.train_steps, "loss": self.last_loss}
