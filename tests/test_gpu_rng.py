"""Per-game counter-based random rows (az_draw_rng; SURVEY.md 8d "per-game RNG = Philox(seed, subsequence = game id)").

What they replace in the reference: one torch generator per worker (training/alphazero_trainer.py:175-177), a Dirichlet row
per move (core/MCTS_alphazero.py:187) and the exponentials behind torch.multinomial (training/self_play.py:73-74).  The
generator is not torch's (parity with torch's stream is unpinned by design; a seeded replay of the reference feeds rows drawn
by torch, tests/test_gpu_arena.py); what is pinned here:
  * the engine's rows == the oracle's restatement (az_oracle_draw_rng) bit for bit, on all four boards;
  * a game's moves and samples depend on (seed, game id) only: the same 30 games through 6 slots, through 15 slots, and as
    2 ranks x 8 slots (gloo, both on cuda:0) are identical game by game, and each replays on the oracle from its rows."""
import functools
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from .conftest import ROOT
from .helpers import HashNet, ListPool

pytestmark = pytest.mark.gpu

BOARDS = [("tictactoe", 0, 3), ("dab", 1, 3), ("gomoku9", 2, 9), ("gomoku15", 2, 15)]


@pytest.mark.parametrize("name,game,n", BOARDS)
def test_engine_rows_equal_the_oracle_rows(oracle, name, game, n):
    from alphazero_amd import Engine

    G, seed = 24, 0x1234_5678_9ABC_DEF1
    eng = Engine(game, G, 8, board_size=n, device="cuda:0")
    A = eng.dims.n_action
    ids_h = torch.tensor([0, 1, 2, 3, 2 ** 31, 2 ** 32 + 5, 2 ** 40 + 7, 2 ** 62] + list(range(100, 100 + G - 8)), dtype=torch.int64)
    ids = ids_h.cuda()
    noise, expo = eng.new(G, A), eng.new(G, A)
    it = torch.ones(G, device="cuda")
    lg, vl = eng.new(G, A), eng.new(G)
    for ply in range(3):  # rows are keyed on the slot's own ply: advance the games and draw again
        noise.fill_(-7.0), expo.fill_(-7.0)
        eng.draw_rng(seed, ids, noise=noise, expo=expo)
        eng.check()
        nh, eh = noise.cpu().numpy(), expo.cpu().numpy()
        for g in range(G):
            rn, re = oracle.draw_rng(seed, int(ids_h[g]), ply, A)
            assert np.array_equal(nh[g], rn) and np.array_equal(eh[g], re), (g, ply)
        assert np.all(eh > 0) and np.all(nh >= 0) and np.allclose(nh.sum(1), 1.0, atol=1e-5)
        # one cheap move of every game
        eng.begin_search()
        for _ in range(2):
            eng.select_gather(None)
            eng.eval_hash(None, lg, vl, 5.0)
            eng.expand_backup(lg, vl, None)
        eng.play_moves(it, expo)
    # another seed, another id: other rows; inactive slots are left alone
    other = eng.new(G, A)
    eng.draw_rng(seed + 1, ids, noise=other)
    assert not torch.equal(other, noise)
    eng.close()


def _games_from_trace(trace, pool_items):
    """-> {game id: (actions, [(state, pi, z) in put order])} of a traced self_play_many run"""
    acts, out, k = {}, {}, 0
    for rnd in trace:
        for s, gid in enumerate(rnd["game_ids"].tolist()):
            a = int(rnd["actions"][s])
            if gid >= 0 and a >= 0:
                acts.setdefault(gid, []).append(a)
        for gid, T in zip(rnd["finished_ids"], rnd["finished_plies"]):
            assert len(acts[gid]) == T
            out[gid] = (acts[gid], [tuple(t.numpy().copy() for t in pool_items[k + j]) for j in range(T)])
            k += T
    assert k == len(pool_items)
    return out


def _same(a, b):
    assert a[0] == b[0]
    assert len(a[1]) == len(b[1])
    for x, y in zip(a[1], b[1]):
        assert all(np.array_equal(p, q) for p, q in zip(x, y))


_RANK_SCRIPT = r"""
import functools, json, os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
from tests.helpers import HashNet, ListPool
from tests.test_gpu_rng import _games_from_trace, _setup
from alphazero_amd.samples import WeightPublisher, games_for_rank
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
arena, tao_of = _setup({name!r}, slots=8, seed={seed})
if rank != 0:  # the learner rank publishes, the others receive: identical weights everywhere before the games start
    with torch.no_grad():
        for p in arena.model.parameters():
            p.add_(1.0)
WeightPublisher(arena.model, src=0).publish()
pool, trace = ListPool(), []
ids = list(games_for_rank({n_games}, rank, world))
arena.self_play_many(len(ids), tao_of, pool, trace=trace, game_ids=ids)
games = _games_from_trace(trace, pool.items)
np.savez({out!r} + f".{{rank}}.npz", **{{f"{{gid}}_a": np.array(g[0]) for gid, g in games.items()}},
         **{{f"{{gid}}_{{j}}_{{k}}": t for gid, g in games.items() for j, s in enumerate(g[1]) for k, t in enumerate(s)}})
dist.barrier()
dist.destroy_process_group()
"""


def _setup(name, slots, seed):
    from alphazero_amd import games, nets
    from alphazero_amd.arena import ChessArena

    torch.manual_seed(5)  # (the networks' initial weights)
    taos = [1.0, 0.8, 0.5]
    tao_of = lambda gid: taos[gid % 3]
    if name == "gomoku9-hashnet":  # the module's own forward (integer-exact network): independent of the batch it runs in
        arena = ChessArena(functools.partial(HashNet, 81, 5.0), games.Gomoku9Game, torch.device("cuda:0"), n_slots=slots,
                           num_simulation=20, seed=seed)
    else:  # the reference's dots-and-boxes network on the one-kernel f32 evaluator (deterministic per board)
        arena = ChessArena(nets.DotsAndBoxesNet, games.DotsAndBoxesGame, torch.device("cuda:0"), n_slots=slots,
                           num_simulation=24, seed=seed, fused=True)
    return arena, tao_of


@pytest.mark.parametrize("name,game,n,sims", [("gomoku9-hashnet", 2, 9, 20), ("dab-fused", 1, 3, 24)])
def test_games_do_not_depend_on_slots_or_ranks(oracle, tmp_path, name, game, n, sims):
    n_games, seed = 30, 4242
    runs = []
    for slots in (6, 15):
        arena, tao_of = _setup(name, slots, seed)
        pool, trace = ListPool(), []
        arena.self_play_many(n_games, tao_of, pool, trace=trace)
        runs.append(_games_from_trace(trace, pool.items))
        model = arena.model
    assert sorted(runs[0]) == list(range(n_games)) == sorted(runs[1])
    for gid in range(n_games):
        _same(runs[0][gid], runs[1][gid])
    assert len({tuple(runs[0][g][0]) for g in range(n_games)}) > n_games // 2  # (the games are not all the same game)

    # 2 ranks x 8 slots, games sharded gid -> rank gid % 2, both ranks on cuda:0, gloo for the weight publish
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "shard")
    script = _RANK_SCRIPT.format(root=ROOT, name=name, seed=seed, n_games=n_games, out=out)
    procs = [subprocess.Popen([sys.executable, "-c", script], cwd=ROOT,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    seen = set()
    for r in range(2):
        z = np.load(out + f".{r}.npz")
        for gid in range(r, n_games, 2):
            acts = z[f"{gid}_a"].tolist()
            samples = [tuple(z[f"{gid}_{j}_{k}"] for k in range(3)) for j in range(len(acts))]
            _same(runs[0][gid], (acts, samples))
            seen.add(gid)
    assert seen == set(range(n_games))

    # and every game is the reference's self-play of that game on the oracle, fed the rows of (seed, game id, ply)
    d = oracle.dims(game, n)
    A, max_plies = d["A"], {0: 9, 1: 24, 2: n * n}[game]
    if name == "gomoku9-hashnet":
        ev = oracle.torch_eval(HashNet(A, 5.0), (d["cin"], d["H"], d["W"]))
    else:
        from alphazero_amd.fused import SmallNetEvaluator

        from .test_gpu_bench_path import _tower_callback

        ev, _calls = _tower_callback(SmallNetEvaluator(model), (d["cin"], d["H"], d["W"]), torch.float32)
    for gid in range(0, n_games, 3):
        acts, samples = runs[0][gid]
        T = len(acts)
        rows = [oracle.draw_rng(seed, gid, t, A) for t in range(T)]
        noise, expo = np.ones((max_plies, A), np.float32), np.ones((max_plies, A), np.float32)
        noise[:T], expo[:T] = np.stack([r[0] for r in rows]), np.stack([r[1] for r in rows])
        r = oracle.self_play(game, n, sims, [1.0, 0.8, 0.5][gid % 3], expo, ev, noise=noise, max_plies=max_plies)
        assert r["plies"] == T and list(r["actions"]) == acts, gid
        for j, (st, pi, zv) in enumerate(samples):
            t = T - 1 - j
            assert np.array_equal(st.reshape(-1), r["nn_states"][t]) and np.array_equal(pi.reshape(-1), r["pis"][t])
            assert float(zv.reshape(-1)[0]) == float(r["z"][t])
