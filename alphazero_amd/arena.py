"""Batched self-play driver: the host-side mirror of the reference's training/self_play.py.

``ChessArena`` keeps the reference's constructor and methods (training/self_play.py:17-98) --
``ChessArena(model_cls, game_cls, device)``, ``.model``, ``.self_play(tao, experience_pool)``,
``.load_model(path)`` -- so ``training/alphazero_trainer.py`` (:182, :213, :368, :464) can use it
unchanged, and adds ``self_play_many`` which plays thousands of games at once on one GPU.

All tree, rule and sample arithmetic runs in libaz_engine.so (HIP); this file only sequences the
calls, runs the policy/value network forward in PyTorch and hands samples to the pool.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Union

import torch

from .engine import Engine, EngineError, GAME_DAB, GAME_GOMOKU, GAME_TICTACTOE

# config/basic.py:12-14, config/train_config.py:5,15-17
C_PUCT = 1.0
ALPHA = 0.3
EPSILON = 0.25
TRAIN_NUM_SIMULATION = 80
START_TEMPERATURE, END_TEMPERATURE, TEMPERATURE_DECAY = 1.0, 0.1, 0.997


def resolve_game(game_cls) -> tuple[int, int]:
    """Map a ``BaseGame`` subclass (the reference's or alphazero_amd.games') to (engine game id, board).

    Only games with a native rule kernel can run on the GPU ("arbitrary user games")."""
    if hasattr(game_cls, "AZ_GAME"):
        return game_cls.AZ_GAME
    name = getattr(game_cls, "__name__", str(game_cls))
    if "TicTacToe" in name:
        return GAME_TICTACTOE, 3
    if "DotsAndBoxes" in name:
        r, c = int(getattr(game_cls, "BOX_ROWS", 3)), int(getattr(game_cls, "BOX_COLS", 3))
        if not (2 <= r <= 5 and 2 <= c <= 5):
            raise EngineError("dots-and-boxes rule kernels cover BOX_ROWS and BOX_COLS 2 .. 5")
        return GAME_DAB, r
    if "Gomoku" in name:
        n, k = int(getattr(game_cls, "BOARD_SIZE", 15)), int(getattr(game_cls, "WIN_LENGTH", 5))
        if not (3 <= n <= 15 and 3 <= k <= 6 and k <= n):
            raise EngineError("gomoku rule kernels cover BOARD_SIZE 3 .. 15 with WIN_LENGTH 3 .. 6")
        return GAME_GOMOKU, n
    raise EngineError(f"no native rule kernel for game class {name}")


def resolve_win_length(game_cls) -> int:
    """the reference's WIN_LENGTH constant of a gomoku class (examples/gomoku/game.py:22); 5 for everything else (ignored)"""
    return int(getattr(game_cls, "WIN_LENGTH", 5)) if "Gomoku" in getattr(game_cls, "__name__", "") or hasattr(game_cls, "WIN_LENGTH") else 5


def resolve_board_cols(game_cls) -> int:
    """the reference's BOX_COLS constant of a dots-and-boxes class (examples/dots_and_boxes/game.py:25); 0 for everything else
    (the engine's board_cols: ignored)"""
    return int(getattr(game_cls, "BOX_COLS", 0)) if hasattr(game_cls, "BOX_ROWS") else 0


def resolve_history_len(game_cls) -> int:
    """the reference's HISTORY_LEN (config/basic.py:10): a ``HISTORY_LEN`` attribute of the class (alphazero_amd.games), else the
    value the class's own module got from ``from config import *`` (every game module of the reference does that), else 8.
    USE_HISTORY = False is refused: the reference's own no-history path does not run (training/self_play.py:59-70 concatenates a
    2-D state with a 3-D player plane; core/MCTS_alphazero.py:208 does the opposite for 3-D states)."""
    import sys

    mod = sys.modules.get(getattr(game_cls, "__module__", ""), None)
    if getattr(mod, "USE_HISTORY", True) is False:
        raise EngineError("USE_HISTORY = False has no native path (and does not run in the reference either)")
    h = getattr(game_cls, "HISTORY_LEN", None)
    if h is None:
        h = getattr(mod, "HISTORY_LEN", 8)
    h = int(h)
    if not 1 <= h <= 8:
        raise EngineError("HISTORY_LEN 1 .. 8")
    return h


def engine_constants(game_cls) -> dict:
    """keyword arguments of ``Engine`` that carry a game class's constants beyond (game id, board size)"""
    return {"win_length": resolve_win_length(game_cls), "board_cols": resolve_board_cols(game_cls),
            "history_len": resolve_history_len(game_cls)}


TaoSchedule = Union[float, Sequence[float], Callable[[int], float]]


def reference_tao(game_index: int) -> float:
    """temperature of a worker's game_index-th game (training/alphazero_trainer.py:186,214)"""
    return max(START_TEMPERATURE * TEMPERATURE_DECAY ** game_index, END_TEMPERATURE)


class SelfPlayRunner:
    """One move of search + play for every active slot of an engine; used by ChessArena and bench.py."""

    def __init__(self, engine: Engine, model: torch.nn.Module, num_sims: int, alpha: float = ALPHA,
                 use_dirichlet: bool = True, reference_rng: bool = False, channels_last: bool = False,
                 fused: bool = False, use_graph: bool = False, graph_unit: Optional[int] = None, seed: Optional[int] = None):
        """seed: when given, the Dirichlet and Exponential rows are the engine's per-game counter-based rows (az_draw_rng:
        Philox keyed on (seed, ``game_ids[slot]``, ply)) -- a game then plays the same moves whatever slot, slot count, rank
        or world size it runs on; the caller keeps ``game_ids`` (int64, one global game id per slot) up to date when it
        restarts slots.  None: rows from torch's device generator (or, with reference_rng, from torch's CPU generator in
        the reference's order: training/alphazero_trainer.py:175-177, core/MCTS_alphazero.py:187, training/self_play.py:73-74).
        graph_unit (with use_graph): simulation steps captured per HIP graph; None = the whole search of a move but its first
        descent and last backup (one replay per move: on the small games a replay per step costs the host more than the step's
        kernels take -- tic-tac-toe +27 %, dots-and-boxes +8 % measured)"""
        if num_sims > engine.max_sims:
            raise EngineError("num_sims exceeds the engine's max_sims")
        self.e, self.model, self.num_sims = engine, model, num_sims
        self.alpha, self.use_dirichlet, self.reference_rng = alpha, use_dirichlet, reference_rng
        self.channels_last = channels_last
        G, A = engine.n_slots, engine.dims.n_action
        dev = engine.device
        self.nn_input = engine.new_nn_input()
        R = engine.rows  # G x leaves per step: rows of the network batch
        self.logits = torch.empty(R, A, device=dev)
        self.values = torch.empty(R, device=dev)
        self.noise = torch.empty(G, A, device=dev)
        self.expo = torch.empty(G, A, device=dev)
        self.actions = torch.empty(G, dtype=torch.int32, device=dev)
        self.done = torch.empty(G, dtype=torch.uint8, device=dev)
        self._conc = torch.full((G, A), alpha, device=dev)
        self.seed = None if seed is None else int(seed)
        self.game_ids = torch.arange(G, dtype=torch.int64, device=dev)  # global game id of every slot (per-game RNG)
        p = next(model.parameters(), None)
        self.model_device = p.device if p is not None else torch.device("cpu")
        self.model_dtype = p.dtype if p is not None else torch.float32
        self.use_graph, self._graph = use_graph, None
        self.graph_unit = max(1, min(int(graph_unit) if graph_unit else num_sims - 1, num_sims - 1)) if num_sims > 1 else 1
        self.fused = None
        if fused:  # hand-written MFMA network kernels (csrc/az_tower*.hip, csrc/az_smallnet.hip) instead of the module's own forward
            from .fused import FusedTowerEvaluator, SmallNetEvaluator

            if engine.game == GAME_GOMOKU:
                if engine.nn_dtype not in (torch.bfloat16, torch.float32):
                    raise EngineError("the fused evaluator exists in bf16 and f32: create the engine with one of these nn_dtypes")
                # the engine's nn_dtype decides: f32 = the reference's own arithmetic, bf16 = throughput mode.  fused=True picks
                #   f32 engine: "f32w" (az_tower_w.hip: f32 MFMAs, the 3x3 convolutions by the Winograd transform -- 1.6x the direct
                #     tower, same accuracy against an f64 evaluation) when the network fits it (15x15, 64 channels), else "f32"
                #     (az_tower_f32.hip: direct convolution);
                #   bf16 engine: "bf16r" (az_tower_h1.hip, the row-tile tower) when the network fits it, else "bf16" (az_tower.hip).
                # By name only: "f16x2" (f32 engine: az_tower_h2.hip, f32-level accuracy on the f16 MFMA pipe, secondary),
                # "f16" (az_tower_h1.hip with f16 operands), and any of the above to override the choice.
                base = "f32" if engine.nn_dtype == torch.float32 else "bf16"
                prec = base
                if not isinstance(fused, str):
                    if base == "bf16" and FusedTowerEvaluator.row_tile_supports(model):
                        prec = "bf16r"  # (the runner feeds the packed batch: no dense dtype involved)
                    elif base == "f32" and FusedTowerEvaluator.winograd_supports(model):
                        prec = "f32w"
                else:
                    if fused not in ("f32", "f32w", "bf16", "f16x2", "f16", "bf16r") or (fused in ("f32", "f32w", "bf16") and fused.rstrip("w") != base):
                        raise EngineError(f"fused={fused!r} does not fit an engine with nn_dtype {engine.nn_dtype}")
                    prec = fused
                self.fused = FusedTowerEvaluator(model, prec)
                if self.fused.board != engine.dims.height:
                    raise EngineError("the fused evaluator's board does not match the engine's game")
            else:  # the reference's tic-tac-toe / dots-and-boxes networks: one f32 kernel per evaluation
                self.fused = SmallNetEvaluator(model)
                if self.fused.game != engine.game or self.fused.n_action != engine.dims.n_action:
                    raise EngineError("the fused evaluator's network does not match the engine's game")
            ring, leaf, _, _, flags = engine.leaf_buffers()
            self._packed = (ring, leaf, None, None, flags)  # flag mode: the tower skips terminal leaves by itself

    # one model forward on the gathered batch -> self.logits / self.values (f32, engine device)
    def evaluate(self) -> None:
        if self.fused is not None:
            _, v = self.fused(None, logits_out=self.logits, packed=self._packed, n_boards=self.e.rows)
            self.values = v.view(-1)  # handed to expand_backup as is (no copy)
            return
        x = self.nn_input
        if x.device != self.model_device or x.dtype != self.model_dtype:
            x = x.to(device=self.model_device, dtype=self.model_dtype)
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.inference_mode():
            p, v = self.model(x)
        self.logits.copy_(p.reshape(self.logits.shape), non_blocking=True)
        self.values.copy_(v.reshape(-1), non_blocking=True)

    def draw_noise(self) -> Optional[torch.Tensor]:
        if not self.use_dirichlet:
            return None
        if self.reference_rng:  # the reference's draw: Dirichlet(ones_like(policy) * ALPHA).sample() on CPU, G == 1
            from torch.distributions import Dirichlet

            A = self.e.dims.n_action
            self.noise.copy_(Dirichlet(torch.ones(self.e.n_slots, A) * self.alpha).sample())
        elif self.seed is not None:
            self.e.draw_rng(self.seed, self.game_ids, noise=self.noise, alpha=self.alpha)
        else:
            self.noise.copy_(torch._sample_dirichlet(self._conc))
        return self.noise

    def draw_expo(self) -> torch.Tensor:
        if self.reference_rng:  # what torch.multinomial(w, 1) draws on CPU (ATen single-sample path)
            self.expo.copy_(torch.empty(self.expo.shape).exponential_(1))
        elif self.seed is not None:  # (same ply as the move's Dirichlet row: drawn before az_play_moves advances it)
            self.e.draw_rng(self.seed, self.game_ids, expo=self.expo)
        else:
            self.expo.exponential_(1)
        return self.expo

    def sim_step(self, noise) -> None:
        """one simulation in every active tree: select+gather -> network -> expand+backup"""
        self.e.select_gather(None if self.fused is not None else self.nn_input)  # fused: packed leaf batch only
        self.evaluate()
        self.e.expand_backup(self.logits, self.values, noise)

    def eval_step(self, noise) -> None:
        """the repeating unit of a search whose first descent is already out: network -> (expand+backup of this
        simulation and select+gather of the next one, in one launch)"""
        self.evaluate()
        self.e.expand_select(self.logits, self.values, noise, None if self.fused is not None else self.nn_input)

    def _capture(self, noise) -> None:
        """Capture the repeating unit into a HIP graph (all buffers are persistent, the launches are plain kernel
        launches on the capture stream): the inner loop is launch-bound for the small games."""
        s = torch.cuda.Stream(self.e.device)
        s.wait_stream(torch.cuda.current_stream(self.e.device))
        with torch.cuda.stream(s):
            self.e.select_gather(None if self.fused is not None else self.nn_input)
            for _ in range(2):  # warm up allocator / library handles outside capture (results are discarded by
                self.eval_step(noise)  # the begin_search that follows)
        torch.cuda.current_stream(self.e.device).wait_stream(s)
        torch.cuda.synchronize(self.e.device)
        g = torch.cuda.CUDAGraph()
        # No garbage collection inside the capture: a stale Engine (or any tensor) finalised there frees device memory,
        # which a stream capture in progress does not survive ("operation failed due to a previous error during capture").
        import gc

        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g):
                for _ in range(self.graph_unit):
                    self.eval_step(noise)
        finally:
            if was_enabled:
                gc.enable()
        self._graph, self._graph_noise = g, noise is not None
        self._graph_gen = getattr(self.fused, "generation", 0)

    def search(self) -> None:
        """num_sims simulations in every active tree (MCTS.search, core/MCTS_alphazero.py:128-178): one descent, then
        num_sims - 1 x (network, backup + next descent), then the last network evaluation and backup"""
        noise = self.draw_noise()
        e = self.e
        if e.leaves_per_step > 1:
            return self._search_waves(noise)
        if self.use_graph and (self._graph is None or self._graph_noise != (noise is not None)
                               or self._graph_gen != getattr(self.fused, "generation", 0)):  # (a refresh() changed a by-value kernel argument)
            self._capture(noise)
        e.begin_search()
        e.select_gather(None if self.fused is not None else self.nn_input)
        if self._graph is not None and self.use_graph:
            for _ in range((self.num_sims - 1) // self.graph_unit):
                self._graph.replay()
            for _ in range((self.num_sims - 1) % self.graph_unit):
                self.eval_step(noise)
        else:
            for _ in range(self.num_sims - 1):
                self.eval_step(noise)
        self.evaluate()
        e.expand_backup(self.logits, self.values, noise)

    def _search_waves(self, noise) -> None:
        """the search with virtual loss / in-tree leaf batching (engine created with leaves_per_step K > 1): every step
        evaluates up to K leaves per tree in one network batch (rows whose flag is 0 -- unused leaves of a tree, trees that
        are done -- cost no evaluation: the fused evaluators skip them).  A tree whose descents collide on a pending leaf
        queues fewer than K, so the number of steps is not known in advance.  No host sync per step: the fewest steps that
        could suffice plus a quarter are queued blind (a tree that is done idles through the rest), then the engine is asked
        ONCE how many simulations the slowest tree still needs, and only a straggler costs further (synchronising) rounds."""
        e, K = self.e, self.e.leaves_per_step
        nn_in = None if self.fused is not None else self.nn_input
        e.begin_search()
        self.wave_steps = self.wave_syncs = 0

        def steps(n):
            for _ in range(n):
                e.select_wave(self.num_sims, nn_in)
                self.evaluate()
                e.expand_wave(self.logits, self.values, noise)
            self.wave_steps += n

        need = (self.num_sims + K - 1) // K
        steps(need + (need + 3) // 4)
        while True:
            _, left = e.search_remaining(self.num_sims)
            self.wave_syncs += 1
            if left <= 0:
                break
            steps(max((left + K - 1) // K, 4))

    def check(self) -> None:
        """the engine's device error flag (Engine.check) and, for the f16 towers, the overflow flag: an activation beyond f16's
        range means every evaluation since is wrong -- raised, never ignored (use the f32 or bf16 tower for such a network).
        Synchronises; call it where the reference would have seen the exception: at the end of a batch of moves."""
        self.e.check()
        if self.fused is not None and getattr(self.fused, "overflowed", lambda: False)():
            raise EngineError(f"the {self.fused.precision} tower overflowed f16's range (an activation above 4,094): its results are "
                              "invalid for this network -- use fused='f32' or 'bf16r'")

    def move_round(self, inv_tao: torch.Tensor, forced: Optional[torch.Tensor] = None):
        """search + temperature sampling + step, for every active slot (training/self_play.py:36-83)"""
        self.search()
        self.draw_expo()
        self.e.play_moves(inv_tao, self.expo, forced, self.actions, self.done)
        return self.actions, self.done


class ChessArena:
    """Drop-in for training/self_play.py:17 ``ChessArena`` backed by the HIP many-tree engine."""

    def __init__(self, model_cls, game_cls, device: Optional[torch.device] = None, *, n_slots: int = 1024,
                 num_simulation: int = TRAIN_NUM_SIMULATION, c_puct: float = C_PUCT, epsilon: float = EPSILON,
                 alpha: float = ALPHA, nn_dtype: torch.dtype = torch.float32, fused: Optional[bool] = False,
                 use_graph: bool = False, seed: Optional[int] = None) -> None:
        """``seed``: seed of the per-game counter-based RNG of ``self_play_many`` (every game's noise and move sampling are
        a function of (seed, game id, ply) only: SelfPlayRunner); None = drawn once from torch's CPU generator, so
        ``torch.manual_seed`` makes a run reproducible (the reference: SEED_BIAS + worker id, alphazero_trainer.py:175-177).
        ``fused``: False (default) = the module's own PyTorch-ROCm forward, whatever the module is -- a seeded run then
        replays a seeded reference run move for move; True = the hand-written network kernels (alphazero_amd.fused: f32 towers
        at nn_dtype float32 -- the reference's arithmetic, agreeing with the module's forward to f32 rounding --, the bf16
        tower at bfloat16; raises if none covers the module); None = use them when one covers the module.  ``use_graph``:
        replay the simulation step as a HIP graph (what the small, launch-bound games want)."""
        if not torch.cuda.is_available():
            raise EngineError("alphazero_amd.ChessArena needs a GPU (no CPU fallback)")
        # the reference defaults to SELF_PLAY_DEVICE = "cpu" (config/train_config.py:11); here the GPU is the point
        self.device = torch.device(device) if device is not None and torch.device(device).type == "cuda" \
            else torch.device("cuda", torch.cuda.current_device())
        self.model = model_cls().to(self.device)
        self.model.eval()
        self.game_cls = game_cls
        self.game_id, self.board_size = resolve_game(game_cls)
        self.win_length, self.board_cols = resolve_win_length(game_cls), resolve_board_cols(game_cls)
        self.history_len = resolve_history_len(game_cls)
        self.num_simulation, self.c_puct, self.epsilon, self.alpha = num_simulation, c_puct, epsilon, alpha
        self.n_slots, self.nn_dtype = n_slots, nn_dtype
        self.fused, self.use_graph = fused, use_graph
        self.seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        self._engines: dict[int, Engine] = {}
        self.games_played = 0

    def _engine(self, slots: int) -> Engine:
        if slots not in self._engines:
            self._engines[slots] = Engine(self.game_id, slots, self.num_simulation, board_size=self.board_size,
                                          c_puct=self.c_puct, epsilon=self.epsilon, nn_dtype=self.nn_dtype,
                                          device=self.device, win_length=self.win_length, board_cols=self.board_cols,
                                          history_len=self.history_len)
        return self._engines[slots]

    # ---- reference surface -------------------------------------------------------------------
    def self_play(self, tao: float, experience_pool) -> None:
        """One game (training/self_play.py:30-91).  Draws its Dirichlet / multinomial randomness from
        torch's global CPU generator in the reference's order, so a seeded run replays the reference."""
        self.self_play_many(1, float(tao), experience_pool, slots=1, reference_rng=True)

    def load_model(self, file_path: str) -> None:
        try:
            with open(file_path, "rb") as f:
                self.model.load_state_dict(torch.load(f, weights_only=True, map_location=self.device))
        except Exception as exc:  # the reference raises FileExistsError here (training/self_play.py:97-98)
            raise FileExistsError(f"model file {file_path} could not be loaded") from exc

    # ---- batched extension --------------------------------------------------------------------
    def self_play_many(self, n_games: int, tao: TaoSchedule, experience_pool, *, slots: Optional[int] = None,
                       reference_rng: bool = False, trace: Optional[list] = None, game_ids: Optional[Sequence[int]] = None) -> dict:
        """Play n_games games, up to ``slots`` at a time; every finished game's samples go to
        ``experience_pool.put_tupule_experience`` exactly as training/self_play.py:110-139 puts them
        ((1,C_in,H,W) f32, (1,A) f32, (1,) f32 CPU tensors, last position first).  Pools that offer
        ``put_batch(states, pis, zs)`` get whole games' samples in one call instead.

        ``trace``: a list that receives one dict per move round -- the Dirichlet and Exponential rows the device drew
        for every slot, 1/tao, the actions played, the slots whose game ended (in the order their samples were put)
        and the slots restarted -- enough to replay every game elsewhere (the parity tests do, on the CPU oracle).

        ``game_ids``: the global ids of the n_games games, in starting order (default: games_played, games_played + 1, ...;
        a rank of a sharded run passes ``samples.games_for_rank``).  A game's random rows, and its temperature when ``tao``
        is a schedule, depend on its id only -- not on the slot, the slot count or the rank that plays it."""
        slots = min(slots or self.n_slots, n_games)
        eng = self._engine(slots)
        fused = self.fused
        if fused is None:
            from .fused import evaluator_available

            fused = evaluator_available(self.game_id, self.model, self.nn_dtype)
        # (a new runner per call: a fused evaluator folds and packs the module's CURRENT weights here, so a load_model /
        # state_dict refresh between calls is picked up)
        run = SelfPlayRunner(eng, self.model, self.num_simulation, alpha=self.alpha, reference_rng=reference_rng, fused=bool(fused),
                             use_graph=self.use_graph, seed=None if reference_rng else self.seed)
        d = eng.dims
        ids = [self.games_played + i for i in range(n_games)] if game_ids is None else [int(i) for i in game_ids]
        if len(ids) != n_games:
            raise EngineError("game_ids must name exactly n_games games")

        def tao_of(i: int) -> float:
            if callable(tao):
                return float(tao(i))
            if isinstance(tao, (int, float)):
                return float(tao)
            return float(tao[i])

        started = min(slots, n_games)
        inv_tao_h = torch.tensor([1.0 / tao_of(ids[i]) if i < started else 1.0 for i in range(slots)], dtype=torch.float32)
        inv_tao = inv_tao_h.to(eng.device)
        ids_h = torch.tensor([ids[i] if i < started else -1 for i in range(slots)], dtype=torch.int64)
        run.game_ids.copy_(ids_h)
        eng.reset_games(None)
        finished = samples = 0
        put_batch = getattr(experience_pool, "put_batch", None)
        ring_writer = None
        if put_batch is None:
            from .samples import SharedRingWriter

            # the reference's shared-memory ring: written in bulk, augmented on the GPU as its own config switches say, straight
            # from the device tensors (one device-to-host copy of the finished rows, none of the raw samples)
            if SharedRingWriter.accepts(experience_pool):
                ring_writer = SharedRingWriter(experience_pool, eng)
        while finished < n_games:
            acts, done = run.move_round(inv_tao)
            done_h = done.cpu()
            idx = torch.nonzero(done_h).reshape(-1)
            if trace is not None:
                trace.append({"noise": run.noise.cpu().clone(), "expo": run.expo.cpu().clone(), "inv_tao": inv_tao.cpu().clone(),
                              "actions": acts.cpu().clone(), "finished": idx.tolist(), "restarted": [], "game_ids": ids_h.clone(),
                              "finished_ids": [], "finished_plies": []})
            if idx.numel() == 0:
                continue
            plies, _, _, _ = eng.game_info()
            plies_h = plies.cpu()[idx]
            if trace is not None:  # which games these are, and how many samples each one puts (in this order)
                trace[-1]["finished_ids"], trace[-1]["finished_plies"] = ids_h[idx].tolist(), plies_h.tolist()
            states, pis, zs, offs = eng.emit_samples(idx.to(torch.int32), plies_h)
            if ring_writer is not None:
                ring_writer.put_batch(states, pis, zs)
            else:  # host tensors, as the reference's pools take them
                states_h = states.view(-1, d.nn_channels, d.height, d.width).cpu()
                pis_h, zs_h = pis.cpu(), zs.cpu()
                if put_batch is not None:
                    put_batch(states_h, pis_h, zs_h)
                else:
                    for k in range(states_h.shape[0]):
                        experience_pool.put_tupule_experience(
                            (states_h[k:k + 1].clone(), pis_h[k:k + 1].clone(), zs_h[k:k + 1].clone()))
            samples += int(states.shape[0])
            finished += int(idx.numel())
            # restart finished slots while games remain (slot recycling)
            restart = torch.zeros(slots, dtype=torch.uint8)
            for s in idx.tolist():
                if started < n_games:
                    restart[s] = 1
                    inv_tao_h[s] = 1.0 / tao_of(ids[started])
                    ids_h[s] = ids[started]
                    started += 1
            if trace is not None:
                trace[-1]["restarted"] = torch.nonzero(restart).reshape(-1).tolist()
            if bool(restart.any()):
                inv_tao.copy_(inv_tao_h)
                run.game_ids.copy_(ids_h)
                eng.reset_games(restart.to(eng.device))
        run.check()
        self.games_played += n_games
        return {"games": n_games, "samples": samples}
