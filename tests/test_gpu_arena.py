"""GPU tests of the drop-in boundary: ChessArena (training/self_play.py surface) and MCTS.search
(core/MCTS_alphazero.py:128) served by the HIP engine, against the CPU oracle driven with the same
seeded randomness and the same (integer-exact) network."""
import functools
import sys

import numpy as np
import pytest
import torch

from .helpers import HashNet, ListPool

pytestmark = pytest.mark.gpu

GAMES = [("tictactoe", 0, 3, 9, 40), ("dab", 1, 3, 24, 30), ("gomoku9", 2, 9, 81, 24)]


def _game_cls(name):
    from alphazero_amd import games

    return {"tictactoe": games.TicTacToeGame, "dab": games.DotsAndBoxesGame, "gomoku9": games.Gomoku9Game}[name]


def _reference_draws(seed, A, plies):
    """the draws a seeded reference game makes, in its order: Dirichlet then multinomial's exponentials"""
    from torch.distributions import Dirichlet

    torch.manual_seed(seed)
    noise, expo = [], []
    for _ in range(plies):
        noise.append(Dirichlet(torch.ones(1, A) * 0.3).sample().reshape(-1).numpy())
        expo.append(torch.empty(1, A).exponential_(1).reshape(-1).numpy())
    return np.array(noise, np.float32), np.array(expo, np.float32)


@pytest.mark.parametrize("name,game,n,A,sims", GAMES)
@pytest.mark.parametrize("tao", [1.0, 0.5])
def test_arena_self_play_replays_like_the_reference(oracle, name, game, n, A, sims, tao):
    from alphazero_amd.arena import ChessArena

    seed = 1234
    max_plies = {0: 9, 1: 24, 2: 81}[game]
    noise, expo = _reference_draws(seed, A, max_plies)
    arena = ChessArena(functools.partial(HashNet, A, 5.0), _game_cls(name), torch.device("cuda:0"), num_simulation=sims)
    assert not arena.model.training
    pool = ListPool()
    torch.manual_seed(seed)
    arena.self_play(tao, pool)
    d = oracle.dims(game, n)
    ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
    r = oracle.self_play(game, n, sims, tao, expo, ev, noise=noise, max_plies=max_plies)
    assert len(pool.items) == r["plies"]
    for k, (st, pi, z) in enumerate(pool.items):  # put order: last position first
        t = r["plies"] - 1 - k
        assert st.shape == (1, d["cin"], d["H"], d["W"]) and st.dtype == torch.float32 and st.device.type == "cpu"
        assert pi.shape == (1, A) and pi.dtype == torch.float32 and z.shape == (1,) and z.dtype == torch.float32
        assert np.array_equal(st.numpy().reshape(-1), r["nn_states"][t])
        assert np.array_equal(pi.numpy().reshape(-1), r["pis"][t])
        assert float(z[0]) == float(r["z"][t])


@pytest.mark.parametrize("n,k", [(7, 4), (12, 4), (6, 3)])
def test_arena_on_a_gomoku_class_with_other_constants(oracle, n, k):
    """the drop-in contract for the reference's GomokuGame with its BOARD_SIZE / WIN_LENGTH constants edited
    (examples/gomoku/game.py:21-22): ChessArena maps the class to az::Gomoku<N> with the win length as a run-time value; a
    seeded game replays on the oracle running the same constants, and the mirror's own static rules agree with the result"""
    from alphazero_amd import games
    from alphazero_amd.arena import ChessArena
    from alphazero_amd.mcts import MCTS

    cls = games.gomoku_game(n, k)
    assert cls.BOARD_SIZE == n and cls.WIN_LENGTH == k
    A, sims, seed = n * n, 30, 77
    noise, expo = _reference_draws(seed, A, A)
    arena = ChessArena(functools.partial(HashNet, A, 5.0), cls, torch.device("cuda:0"), num_simulation=sims)
    assert (arena.board_size, arena.win_length) == (n, k)
    pool = ListPool()
    torch.manual_seed(seed)
    arena.self_play(1.0, pool)
    with oracle.win_length(k):
        d = oracle.dims(2, n)
        ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
        r = oracle.self_play(2, n, sims, 1.0, expo, ev, noise=noise, max_plies=A)
    assert len(pool.items) == r["plies"]
    for j, (st, pi, z) in enumerate(pool.items):
        t = r["plies"] - 1 - j
        assert st.shape == (1, 9, n, n) and np.array_equal(st.numpy().reshape(-1), r["nn_states"][t])
        assert np.array_equal(pi.numpy().reshape(-1), r["pis"][t]) and float(z[0]) == float(r["z"][t])
    # the final board through the mirror's own rules (games.GomokuGame._get_winner is generic in both constants)
    last_state = pool.items[0][0][0, 7].clone()
    a_last = int(r["actions"][-1])
    mover = 1 if (r["plies"] - 1) % 2 == 0 else -1
    final, _ = cls.next_state(last_state, a_last, mover)
    assert cls.is_terminal(final)
    w = cls._get_winner(final)
    assert (w is None) == (r["result"] == 0.0) and (w is None or w == mover)
    # and the single-position adapter
    g = cls()
    g.step(n + 1)
    pi = MCTS(cls).search(HashNet(A, 5.0).cuda(), g.get_state(), g.get_player(), history_state=g.get_history(), num_simulation=20,
                          use_Dirichlet=False)
    assert pi.shape == (1, A) and abs(float(pi.sum()) - 1.0) < 1e-5 and float(pi[0, n + 1]) == 0.0


@pytest.mark.parametrize("rows,cols", [(2, 2), (4, 3), (5, 5)])
def test_arena_on_a_dots_and_boxes_class_with_other_constants(oracle, rows, cols):
    """the same for the reference's DotsAndBoxesGame with BOX_ROWS / BOX_COLS edited (examples/dots_and_boxes/game.py:24-25):
    ChessArena maps the class to az::Dab<R, C>; a seeded game replays on the oracle under the same constants, the mirror's own
    rules replay it move for move, and the mirror network (which, like the reference's, sizes itself from the constants) runs"""
    from alphazero_amd import games, nets
    from alphazero_amd.arena import ChessArena
    from alphazero_amd.mcts import MCTS

    cls = games.dots_and_boxes_game(rows, cols)
    A, sims, seed = (rows + 1) * cols + rows * (cols + 1), 30, 78
    assert (cls.BOX_ROWS, cls.BOX_COLS, cls.NUM_ACTION) == (rows, cols, A)
    noise, expo = _reference_draws(seed, A, A)
    arena = ChessArena(functools.partial(HashNet, A, 5.0), cls, torch.device("cuda:0"), num_simulation=sims)
    assert (arena.board_size, arena.board_cols) == (rows, cols)
    pool = ListPool()
    torch.manual_seed(seed)
    arena.self_play(1.0, pool)
    with oracle.box_shape(rows, cols):
        d = oracle.dims(1, rows)
        ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
        r = oracle.self_play(1, rows, sims, 1.0, expo, ev, noise=noise, max_plies=A)
    assert len(pool.items) == r["plies"] == A
    for j, (st, pi, z) in enumerate(pool.items):
        t = r["plies"] - 1 - j
        assert st.shape == (1, 25, rows + 1, cols + 1) and np.array_equal(st.numpy().reshape(-1), r["nn_states"][t])
        assert np.array_equal(pi.numpy().reshape(-1), r["pis"][t]) and float(z[0]) == float(r["z"][t])
    # the same game through the mirror's own rules: every recorded position, the extra turns, the result
    g = cls()
    for t, a in enumerate(r["actions"]):
        assert np.array_equal(g.get_state().numpy().reshape(-1), r["nn_states"][t].reshape(25, -1)[21:24].reshape(-1))
        assert g.get_player() == int(r["nn_states"][t].reshape(25, -1)[24, 0])
        over = g.step(int(a))
        assert over == (t == A - 1)
    w = cls._get_winner(g.get_state())
    assert (w is None) == (r["result"] == 0.0)
    # the single-position adapter with the mirror network sized from the constants
    net = nets.DotsAndBoxesNet(rows, cols).cuda().eval()
    g = cls()
    g.step(1)
    pi = MCTS(cls).search(net, g.get_state(), g.get_player(), history_state=g.get_history(), num_simulation=20, use_Dirichlet=False)
    assert pi.shape == (1, A) and abs(float(pi.sum()) - 1.0) < 1e-5 and float(pi[0, 1]) == 0.0


@pytest.mark.parametrize("name,h", [("tictactoe", 3), ("dab", 1), ("gomoku9", 4)])
def test_arena_with_another_history_length(oracle, name, h):
    """the reference's HISTORY_LEN constant (config/basic.py:10) at another value: ChessArena reads it from the game class (the
    mirror's attribute; for the reference's own classes, the binding their module got from ``from config import *``), the samples
    have h * C + 1 planes and replay on the oracle under the same constant; the mirror networks take the same argument"""
    import types

    from alphazero_amd import games, nets
    from alphazero_amd.arena import ChessArena, resolve_history_len
    from alphazero_amd.mcts import MCTS

    base = _game_cls(name)
    cls = games.with_history_len(base, h)
    game, n = cls.AZ_GAME
    assert cls.HISTORY_LEN == h and cls().get_history().shape[0] == h * (3 if name == "dab" else 1)
    # a class of the reference's has no such attribute: its module's HISTORY_LEN is what counts
    mod = types.ModuleType("fake_reference_game_module")
    mod.HISTORY_LEN, mod.USE_HISTORY = h, True
    sys.modules[mod.__name__] = mod
    try:
        ref_like = type("GomokuGame", (), {"__module__": mod.__name__, "BOARD_SIZE": 9})
        assert resolve_history_len(ref_like) == h
        mod.USE_HISTORY = False
        with pytest.raises(Exception, match="USE_HISTORY"):
            resolve_history_len(ref_like)
    finally:
        del sys.modules[mod.__name__]
    with oracle.history_len(h):
        d = oracle.dims(game, n)
        A, sims, seed = d["A"], 25, 79
        P = min(A, 30)
        noise, expo = _reference_draws(seed, A, P)
        arena = ChessArena(functools.partial(HashNet, A, 5.0), cls, torch.device("cuda:0"), num_simulation=sims)
        assert arena.history_len == h
        pool = ListPool()
        torch.manual_seed(seed)
        if name != "gomoku9":  # (a 25-simulation gomoku game outlasts the recorded draws)
            arena.self_play(1.0, pool)
            ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
            r = oracle.self_play(game, n, sims, 1.0, expo, ev, noise=noise, max_plies=P)
            assert len(pool.items) == r["plies"]
            for j, (st, pi, z) in enumerate(pool.items):
                t = r["plies"] - 1 - j
                assert st.shape == (1, d["cin"], d["H"], d["W"]) and np.array_equal(st.numpy().reshape(-1), r["nn_states"][t])
                assert np.array_equal(pi.numpy().reshape(-1), r["pis"][t]) and float(z[0]) == float(r["z"][t])
    net = {"tictactoe": lambda: nets.TicTacToeNet(history_len=h), "dab": lambda: nets.DotsAndBoxesNet(history_len=h),
           "gomoku9": lambda: nets.GomokuNet(9, history_len=h)}[name]().cuda().eval()
    g = cls()
    g.step(1)
    pi = MCTS(cls).search(net, g.get_state(), g.get_player(), history_state=g.get_history(), num_simulation=20, use_Dirichlet=False)
    assert pi.shape == (1, A) and abs(float(pi.sum()) - 1.0) < 1e-5 and float(pi[0, 1]) == 0.0


@pytest.mark.parametrize("name,game,n,A,sims", GAMES)
def test_mcts_search_adapter(oracle, name, game, n, A, sims):
    from alphazero_amd.mcts import MCTS
    from torch.distributions import Dirichlet

    cls = _game_cls(name)
    g = cls()
    rng = np.random.default_rng(3)
    for _ in range(4):
        legal = np.nonzero(g.get_legal_mask().numpy().reshape(-1))[0]
        g.step(int(rng.choice(legal)))
    model = HashNet(A, 5.0).cuda()
    mcts = MCTS(cls)
    d = oracle.dims(game, n)
    ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
    for use_dir in (True, False):
        torch.manual_seed(77)
        pi = mcts.search(model, g.get_state(), g.get_player(), history_state=g.get_history(), num_simulation=sims,
                         use_Dirichlet=use_dir)
        assert pi.shape == (1, A) and pi.dtype == torch.float32
        torch.manual_seed(77)
        noise = Dirichlet(torch.ones(1, A) * 0.3).sample().numpy().reshape(-1) if use_dir else None
        r = oracle.search(game, n, g.get_state().numpy(), g.get_player(), g.get_history().numpy(), sims, ev, noise=noise)
        assert np.array_equal(pi.cpu().numpy().reshape(-1), r["pi"])
    with pytest.raises(ValueError):
        mcts.search(model, g.get_state(), g.get_player())


def test_self_play_many_recycles_slots_and_fills_the_pool():
    from alphazero_amd.arena import ChessArena, reference_tao
    from alphazero_amd.games import TicTacToeGame
    from alphazero_amd.nets import TicTacToeNet

    torch.manual_seed(0)
    arena = ChessArena(TicTacToeNet, TicTacToeGame, torch.device("cuda:0"), n_slots=16, num_simulation=30)
    pool = ListPool()
    out = arena.self_play_many(50, reference_tao, pool)  # 50 games through 16 slots
    assert out["games"] == 50 and out["samples"] == len(pool.items) and 5 * 50 <= len(pool.items) <= 9 * 50
    zs = torch.cat([e[2] for e in pool.items])
    assert set(zs.tolist()) <= {-1.0, 0.0, 1.0}
    pis = torch.cat([e[1] for e in pool.items])
    assert torch.allclose(pis.sum(1), torch.ones(len(pool.items)), atol=1e-5)
    st = torch.cat([e[0] for e in pool.items])
    assert st.shape[1:] == (9, 3, 3) and set(st.unique().tolist()) <= {-1.0, 0.0, 1.0}
    # the newest history plane of a sample holds as many stones as plies were played before it; the
    # player plane is +1 exactly when that count is even
    stones = st[:, 7].abs().sum((1, 2))
    assert torch.equal((stones % 2 == 0), (st[:, 8, 0, 0] == 1))
    assert arena.games_played == 50

    class BatchPool:
        def __init__(self):
            self.n = 0

        def put_batch(self, s, p, z):
            assert s.shape[0] == p.shape[0] == z.shape[0] and s.shape[1:] == (9, 3, 3)
            self.n += s.shape[0]

    bp = BatchPool()
    assert arena.self_play_many(20, 1.0, bp)["samples"] == bp.n


def test_device_ring_and_learner_close_the_loop():
    """self-play -> on-device augmentation -> HBM ring -> sample -> the reference's loss, on the GPU"""
    from alphazero_amd.arena import ChessArena
    from alphazero_amd.games import TicTacToeGame
    from alphazero_amd.learner import train_step
    from alphazero_amd.nets import TicTacToeNet
    from alphazero_amd.samples import DeviceSampleRing

    torch.manual_seed(0)
    arena = ChessArena(TicTacToeNet, TicTacToeGame, torch.device("cuda:0"), n_slots=64, num_simulation=20)
    eng = arena._engine(64)
    ring = DeviceSampleRing(eng, capacity=3000)

    class Sink:  # what self_play_many feeds: whole finished games, host tensors
        def put_batch(self, s, p, z):
            ring.put_games(s.reshape(s.shape[0], -1).cuda(), p.cuda(), z.cuda())

    out = arena.self_play_many(128, 1.0, Sink())
    assert ring.total_written == 8 * out["samples"] and ring.size() == min(3000, 8 * out["samples"])
    s, p, z = ring.sample(32)
    assert s.shape == (32, 9, 3, 3) and p.shape == (32, 9) and z.shape == (32, 1) and s.is_cuda
    assert torch.allclose(p.sum(1), torch.ones(32, device="cuda"), atol=1e-5)
    opt = torch.optim.Adam(arena.model.parameters(), lr=1e-3)
    losses = [train_step(arena.model, opt, ring, 64)[0] for _ in range(60)]
    assert sum(losses[-10:]) < sum(losses[:10])  # it learns
    assert not arena.model.training


def test_gpu_worker_has_the_reference_worker_signature():
    import threading

    from alphazero_amd.games import DotsAndBoxesGame
    from alphazero_amd.nets import DotsAndBoxesNet
    from alphazero_amd.worker import gpu_self_player_worker

    torch.manual_seed(3)
    shared = {k: v.clone() for k, v in DotsAndBoxesNet().state_dict().items()}
    pool = ListPool()
    gpu_self_player_worker(DotsAndBoxesNet, DotsAndBoxesGame, shared, pool, 12, 0, threading.Lock(), n_slots=8,
                           num_simulation=10)
    assert len(pool.items) == 12 * 24  # every dots-and-boxes game has exactly 24 plies
    st, pi, z = pool.items[0]
    assert st.shape == (1, 25, 4, 4) and pi.shape == (1, 24) and z.shape == (1,)


def test_self_play_train_loop_runs_and_learns():
    """BASELINE config 5 in miniature: self-play -> device ring -> learner -> refreshed self-play net, one GPU"""
    from alphazero_amd.loop import SelfPlayTrainLoop
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    loop = SelfPlayTrainLoop(GomokuNet(9, 64, 1), 2, 9, 64, 8, torch.device("cuda:0"), ring_capacity=1 << 14, batch_size=128,
                             train_steps_per_move=2, min_samples=512, publish_every=4)
    assert loop.runner.fused is not None  # the bf16 self-play copy runs on the fused tower
    before = {k: v.clone() for k, v in loop.player.state_dict().items()}
    out = loop.run(60)
    assert out["games"] > 0 and out["train_steps"] >= 20 and out["samples"] > 0 and loop.ring.size() > 0
    changed = any(not torch.equal(before[k], v) for k, v in loop.player.state_dict().items())
    assert changed  # the self-play network follows the learner ...
    x = torch.randint(-1, 2, (32, 9, 9, 9), device="cuda").float()
    with torch.inference_mode():
        want_p, want_v = loop.master.eval()(x)
    got_p, got_v = loop.runner.fused(x.bfloat16().contiguous())
    scale = float(want_p.abs().max())
    assert float((got_p - want_p).abs().max()) <= 0.03 * scale + 0.02  # ... and so does the fused tower's folded copy
    assert float((got_v - want_v).abs().max()) <= 0.03

    class Fixed:  # the learner itself: repeated steps on one batch of the loop's own samples must fit it
        batch = loop.ring.sample(256)

        def sample(self, n, timeout=None):
            return self.batch

    losses = [loop.learner.step(Fixed(), 256)[0] for _ in range(40)]
    assert losses[-1] < 0.8 * losses[0]


def test_copy_to_host_moves_every_byte():
    """az_copy_to_host (the few-workgroup D2H the sample path uses) == Tensor.cpu(), odd sizes included."""
    from alphazero_amd.engine import EngineError, copy_to_host

    g = torch.Generator(device="cuda").manual_seed(3)
    for n, dt in ((1, torch.int8), (15, torch.int8), (16, torch.int8), (2025 * 37 + 5, torch.int8), (225 * 1001, torch.float32), (0, torch.float32)):
        src = torch.randint(-100, 100, (n,), device="cuda", generator=g).to(dt)
        dst = torch.zeros(n, dtype=dt).pin_memory()
        copy_to_host(dst, src)
        torch.cuda.synchronize()
        assert torch.equal(dst, src.cpu())
    with pytest.raises(EngineError):
        copy_to_host(torch.zeros(4), torch.zeros(4, device="cuda"))  # not pinned


def _replay_trace(oracle, trace, pool_items, game, n, A, sims, slots, tao_of_game, max_plies, ev=None):
    """Rebuild every game of a traced self_play_many run (per-slot draws of every move round, recycling included) and
    replay it on the CPU oracle: actions, the samples the pool received (put order: finished games in slot order per
    round, last position first) and z must agree bit for bit.  training/self_play.py:30-139."""
    d = oracle.dims(game, n)
    if ev is None:
        ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
    cur = {s: {"noise": [], "expo": [], "actions": [], "game": s} for s in range(slots)}
    next_game = slots
    k = 0  # cursor into pool_items
    games_checked = 0
    for rnd in trace:
        for s, g in cur.items():
            if g is None:
                continue
            a = int(rnd["actions"][s])
            if a < 0:
                continue
            g["noise"].append(rnd["noise"][s].numpy())
            g["expo"].append(rnd["expo"][s].numpy())
            g["actions"].append(a)
            g["inv_tao"] = float(rnd["inv_tao"][s])
        for s in rnd["finished"]:
            g = cur[s]
            T = len(g["actions"])
            tao = tao_of_game(g["game"])
            assert np.float32(1.0 / tao) == np.float32(g["inv_tao"])
            expo = np.zeros((max_plies, A), np.float32)
            noise = np.zeros((max_plies, A), np.float32)
            expo[:T], noise[:T] = np.stack(g["expo"]), np.stack(g["noise"])
            expo[T:] = 1.0
            r = oracle.self_play(game, n, sims, tao, expo, ev, noise=noise, max_plies=max_plies)
            assert r["plies"] == T and list(r["actions"]) == g["actions"], (s, g["game"])
            for j in range(T):  # put order: last position first
                st, pi, z = pool_items[k + j]
                t = T - 1 - j
                assert np.array_equal(st.numpy().reshape(-1), r["nn_states"][t])
                assert np.array_equal(pi.numpy().reshape(-1), r["pis"][t])
                assert float(z.reshape(-1)[0]) == float(r["z"][t])
            k += T
            games_checked += 1
            cur[s] = None
        for s in rnd["restarted"]:
            cur[s] = {"noise": [], "expo": [], "actions": [], "game": next_game}
            next_game += 1
    assert k == len(pool_items)
    return games_checked


@pytest.mark.parametrize("name,game,n,A,sims", GAMES)
def test_self_play_many_with_recycling_replays_on_the_oracle(oracle, name, game, n, A, sims):
    """multi-slot self-play with slot recycling (device-drawn Dirichlet / exponential rows captured per move round)
    == oracle.self_play game by game: actions, samples, z"""
    from alphazero_amd.arena import ChessArena

    slots, n_games = 6, 15
    max_plies = {0: 9, 1: 24, 2: 81}[game]
    torch.manual_seed(11)
    arena = ChessArena(functools.partial(HashNet, A, 5.0), _game_cls(name), torch.device("cuda:0"), n_slots=slots,
                       num_simulation=sims)
    pool, trace = ListPool(), []
    taos = [1.0, 0.8, 0.5]
    tao_of = lambda i: taos[i % 3]
    arena.self_play_many(n_games, tao_of, pool, trace=trace)
    assert _replay_trace(oracle, trace, pool.items, game, n, A, sims, slots, tao_of, max_plies) == n_games


def test_gpu_worker_replays_on_the_oracle(oracle):
    """f-1: the same comparison through gpu_self_player_worker (the reference worker's signature and temperature
    schedule, training/alphazero_trainer.py:164-221)"""
    import threading

    from alphazero_amd.arena import reference_tao
    from alphazero_amd.games import DotsAndBoxesGame
    from alphazero_amd.worker import gpu_self_player_worker

    A, sims, slots, n_games = 24, 16, 4, 9
    model_cls = functools.partial(HashNet, A, 5.0)
    shared = {k: v.clone() for k, v in model_cls().state_dict().items()}
    pool, trace = ListPool(), []
    gpu_self_player_worker(model_cls, DotsAndBoxesGame, shared, pool, n_games, 0, threading.Lock(), n_slots=slots,
                           num_simulation=sims, trace=trace)
    tao_of = lambda i: reference_tao(i // slots)
    assert _replay_trace(oracle, trace, pool.items, 1, 3, A, sims, slots, tao_of, 24) == n_games


@pytest.mark.parametrize("name", ["dab", "gomoku9-quick"])
def test_gpu_worker_with_the_fused_f32_networks_replays_on_the_oracle(oracle, name):
    """f-1 on the fast path: gpu_self_player_worker(fused=None) picks the hand-written f32 network kernel for the reference's
    dots-and-boxes network (HIP-graph replay of the simulation step) and for a QuickModel-family gomoku network; every game
    is replayed on the oracle with that same kernel as the oracle's evaluator (one dense position per call): actions,
    samples, z bit for bit."""
    import threading

    from alphazero_amd import nets
    from alphazero_amd.arena import reference_tao
    from alphazero_amd.fused import FusedTowerEvaluator, SmallNetEvaluator
    from alphazero_amd.games import DotsAndBoxesGame, Gomoku9Game
    from alphazero_amd.worker import gpu_self_player_worker

    from .test_gpu_bench_path import _tower_callback
    from .test_gpu_tower import _randomised

    torch.manual_seed(21)
    if name == "dab":
        game, n, A, sims, slots, n_games, max_plies, game_cls = 1, 3, 24, 16, 5, 11, 24, DotsAndBoxesGame
        proto = nets.DotsAndBoxesNet().eval()
        model_cls = nets.DotsAndBoxesNet
        make_ev = SmallNetEvaluator
    else:
        game, n, A, sims, slots, n_games, max_plies, game_cls = 2, 9, 81, 12, 4, 6, 81, Gomoku9Game
        proto = _randomised(nets.QuickNet(9, 9, 64, 1, 2)).eval()
        model_cls = functools.partial(nets.QuickNet, 9, 9, 64, 1, 2)
        make_ev = lambda m: FusedTowerEvaluator(m, "f32")
    shared = {k: v.clone() for k, v in proto.state_dict().items()}
    pool, trace = ListPool(), []
    gpu_self_player_worker(model_cls, game_cls, shared, pool, n_games, 0, threading.Lock(), n_slots=slots, num_simulation=sims,
                           trace=trace, fused=None, use_graph=(name == "dab"))
    d = oracle.dims(game, n)
    cb, calls = _tower_callback(make_ev(proto.cuda()), (d["cin"], d["H"], d["W"]), torch.float32)
    tao_of = lambda i: reference_tao(i // slots)
    assert _replay_trace(oracle, trace, pool.items, game, n, A, sims, slots, tao_of, max_plies, ev=cb) == n_games
    assert calls[0] > 0


class _RefLikeRing:
    """The fields and the ``put`` / ``put_tupule_experience`` semantics of utils/share_ring_buffer.py:39-127, restated
    (the reference itself does not travel to the GPU box); augmentation through oracle/augment.py, which is pinned to
    the reference's get_enhanced_data by tests/golden/augment.npz."""

    def __init__(self, cap, nn_shape, A, game):
        import multiprocessing as mp

        self._states = torch.zeros((cap, 1, *nn_shape))
        self._prior = torch.zeros((cap, 1, A))
        self._values = torch.zeros((cap, 1))
        self._write_idx, self._read_idx = mp.Value("i", 0), mp.Value("i", 0)
        self._size, self._total_written = mp.Value("i", 0), mp.Value("i", 0)
        self._lock = mp.Lock()
        self._not_empty = mp.Condition(self._lock)
        self._capacity, self._game = cap, game

    def put(self, state, prior, value):
        with self._lock:
            if self._size.value >= self._capacity:
                self._read_idx.value = (self._read_idx.value + 1) % self._capacity
                self._size.value = self._capacity
            idx = self._write_idx.value
            self._states[idx].copy_(state)
            self._prior[idx].copy_(prior)
            self._values[idx].copy_(value)
            self._write_idx.value = (idx + 1) % self._capacity
            self._size.value = min(self._size.value + 1, self._capacity)
            self._total_written.value += 1
            self._not_empty.notify()

    def put_tupule_experience(self, exp):
        from oracle import augment as AUG

        st, pi, z = exp
        outs, outp, outz = AUG.augment(self._game, st.numpy()[0], pi.numpy()[0], float(z[0]))
        for t in range(8):
            self.put(torch.from_numpy(np.ascontiguousarray(outs[t]))[None], torch.from_numpy(np.ascontiguousarray(outp[t]))[None],
                     torch.tensor([outz[t]]))


@pytest.mark.parametrize("name,game,n,A,sims", GAMES)
def test_bulk_writer_fills_the_reference_ring_like_its_own_puts(name, game, n, A, sims):
    """self_play_many -> SharedRingWriter (GPU augmentation, one locked bulk write) leaves the reference-style shared
    ring exactly as the reference's per-sample put_tupule_experience loop would: rows, wrap-around, indices."""
    from alphazero_amd.arena import ChessArena

    torch.manual_seed(5)
    arena = ChessArena(functools.partial(HashNet, A, 5.0), _game_cls(name), torch.device("cuda:0"), n_slots=8, num_simulation=sims)
    d = arena._engine(8).dims
    shape = (d.nn_channels, d.height, d.width)
    listed = ListPool()
    trace = []
    torch.manual_seed(6)
    torch.cuda.manual_seed(6)
    arena.self_play_many(20, 1.0, listed, trace=trace)  # the per-sample path: what the samples are
    cap = 8 * len(listed.items) - 37  # smaller than the data: wraps and overwrites
    want = _RefLikeRing(cap, shape, A, game)
    for exp in listed.items:
        want.put_tupule_experience(exp)
    got = _RefLikeRing(cap, shape, A, game)
    torch.manual_seed(6)
    torch.cuda.manual_seed(6)
    arena.games_played = 0
    arena.self_play_many(20, 1.0, got)  # same seeds => same games, through the bulk writer
    for f in ("_write_idx", "_read_idx", "_size", "_total_written"):
        assert getattr(got, f).value == getattr(want, f).value, f
    assert torch.equal(got._states, want._states) and torch.equal(got._prior, want._prior) and torch.equal(got._values, want._values)


RING = None


def _ring_cases():
    global RING
    if RING is None:
        from .conftest import load_cases

        RING = load_cases("ring.npz")
    return RING


@pytest.mark.parametrize("tag,game,n", [("tictactoe", 0, 3), ("dab", 1, 3), ("gomoku9", 2, 9)])
def test_bulk_writer_against_the_references_own_ring(tag, game, n):
    """tests/golden/ring.npz: the reference's SharedRingBufferExperiencePool itself (utils/share_ring_buffer.py:22-127), fed
    fixed samples through put_tupule_experience in two instalments with a capacity smaller than the data.  SharedRingWriter
    (GPU augmentation + one locked bulk write per instalment) must leave the same rows and the same write / read / size /
    total indices after each instalment -- wrap-around, overwriting the oldest rows, more than one lap included."""
    from alphazero_amd import Engine
    from alphazero_amd.samples import SharedRingWriter

    c = _ring_cases()[tag]
    eng = Engine(game, 4, 4, board_size=n, device="cuda:0")
    d = eng.dims
    cap, split = int(c["capacity"]), int(c["split"])
    ring = _RefLikeRing(cap, (d.nn_channels, d.height, d.width), d.n_action, game)  # (used as the bare fields here)
    w = SharedRingWriter(ring, eng)
    st = torch.from_numpy(c["states"].astype(np.float32))
    pi, z = torch.from_numpy(c["pis"]), torch.from_numpy(c["z"])
    for part, (lo, hi) in enumerate(((0, split), (split, st.shape[0]))):
        assert w.put_batch(st[lo:hi], pi[lo:hi], z[lo:hi]) == 8 * (hi - lo)
        idx = [ring._write_idx.value, ring._read_idx.value, ring._size.value, ring._total_written.value]
        assert idx == c[f"idx_{part}"].tolist(), (part, idx)
        assert np.array_equal(ring._states.numpy().astype(np.int8), c[f"ring_states_{part}"]), part
        assert np.array_equal(ring._prior.numpy(), c[f"ring_prior_{part}"]), part
        assert np.array_equal(ring._values.numpy(), c[f"ring_values_{part}"]), part
    eng.close()


def test_bulk_writer_honours_the_references_augmentation_switches():
    """utils/data_enhancer.py:29-45 under config/utils_config.py's switches: rotation + flip = the 8 rows (pinned above),
    rotation only = the four rotations, flip only = the flipped sample ALONE, enhancement off = the sample itself.  The mode
    comes from the switches the pool's own module holds (as its put_tupule_experience would read them); device tensors in."""
    import sys
    import types

    from alphazero_amd import Engine
    from alphazero_amd.samples import SharedRingWriter

    c = _ring_cases()["gomoku9"]
    eng = Engine(2, 4, 4, board_size=9, device="cuda:0")
    d = eng.dims
    st = torch.from_numpy(c["states"].astype(np.float32))[:5].cuda()
    pi, z = torch.from_numpy(c["pis"])[:5].cuda(), torch.from_numpy(c["z"])[:5].cuda()
    full = _RefLikeRing(64, (d.nn_channels, d.height, d.width), d.n_action, 2)
    assert SharedRingWriter(full, eng).mode == "rot+flip" and SharedRingWriter(full, eng).put_batch(st, pi, z) == 40
    mod = types.ModuleType("fake_share_ring_buffer")
    sys.modules[mod.__name__] = mod
    ring_cls = type("Ring", (_RefLikeRing,), {"__module__": mod.__name__})
    try:
        for flags, mode, rows in (((True, True, False), "rot", (0, 2, 4, 6)), ((True, False, True), "flip", (1,)),
                                  ((False, True, True), "none", (0,)), ((True, False, False), "none", (0,))):
            mod.USE_DATA_ENHANCEMENT, mod.USE_ROATATION, mod.USE_FLIP = flags
            ring = ring_cls(64, (d.nn_channels, d.height, d.width), d.n_action, 2)
            w = SharedRingWriter(ring, eng)
            assert w.mode == mode
            assert w.put_batch(st, pi, z) == 5 * len(rows)
            sel = [8 * i + r for i in range(5) for r in rows]
            k = len(sel)
            assert ring._write_idx.value == k and ring._size.value == k and ring._total_written.value == k
            assert torch.equal(ring._states[:k], full._states[sel]) and torch.equal(ring._prior[:k], full._prior[sel])
            assert torch.equal(ring._values[:k], full._values[sel])
    finally:
        del sys.modules[mod.__name__]
    eng.close()


@pytest.mark.parametrize("kind", ["gomokunet-128x4-fused-bf16", "quicknet-128x2x3-fused-f32", "quicknet-128x2x3-fused-bf16",
                                  "quicknet-128x2x3-pytorch-f32"])
def test_config5_shape_self_play_train_loop(kind):
    """BASELINE config 5 at its real board and width (gomoku 15x15, 128 channels, >= 4 residual conv pairs), a few moves of
    the self-play + training loop: the GomokuNet-family tower on the fused bf16 kernel, and the reference's QuickModel
    architecture (nets.QuickNet, 'quick_model scaled up') on the fused f32 tower (the default for it) and through its own
    PyTorch forward in f32."""
    from alphazero_amd.loop import SelfPlayTrainLoop
    from alphazero_amd.nets import GomokuNet, QuickNet

    torch.manual_seed(0)
    if kind.startswith("gomokunet"):
        net, dt = GomokuNet(15, 128, 4), torch.bfloat16
    else:
        net, dt = QuickNet(15, 9, 128, 2, 3), (torch.bfloat16 if kind.endswith("bf16") else torch.float32)
    loop = SelfPlayTrainLoop(net, 2, 15, 48, 12, torch.device("cuda:0"), ring_capacity=1 << 13, batch_size=64,
                             train_steps_per_move=1, min_samples=256, publish_every=2, nn_dtype=dt,
                             fused=False if "pytorch" in kind else None)
    assert (loop.runner.fused is not None) == ("fused" in kind)
    # a trained-looking value head ends games early; with random weights games run long, so force short games: play
    # until some finish (12 simulations per move, 48 games)
    out = loop.run(14)
    st = loop.engine.read_stats()
    assert st["sims"] == 48 * 12 * 14 and out["moves"] == 14
    # feed the ring by hand from the positions played so far if no game has ended yet (five in a row needs >= 9 plies)
    if loop.ring.size() < 256:
        plies, _, active, _ = loop.engine.game_info()
        assert int(plies.min()) >= 1
        d = loop.engine.dims
        s = loop.engine.root_nn_state().reshape(48, -1)
        pi = loop.engine.root_policy()
        loop.ring.put_games(s, torch.softmax(torch.randn(48, d.n_action, device="cuda"), 1), torch.zeros(48, device="cuda"))
    before = [p.detach().clone() for p in loop.master.parameters()]
    loss = [loop.learner.step(loop.ring, 64)[0] for _ in range(3)]
    assert all(np.isfinite(loss))
    assert any(not torch.equal(a, b) for a, b in zip(before, loop.master.parameters()))
    loop._refresh_player()
    loop.runner.move_round(loop.inv_tao)  # the refreshed network (re-folded tower) still searches
    loop.engine.check()
