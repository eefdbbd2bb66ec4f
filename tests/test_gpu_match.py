"""GPU test of the batched evaluation match (alphazero_amd/match.py) against the CPU oracle:
the same games replayed move by move with oracle.search (no noise) + argmax + oracle.next_state, bit-exact."""
import numpy as np
import pytest
import torch

from .helpers import HashNet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,game,n,A,sims", [("tictactoe", 0, 3, 9, 30), ("dab", 1, 3, 24, 24), ("gomoku9", 2, 9, 81, 20)])
def test_match_replays_like_the_oracle(oracle, name, game, n, A, sims):
    from alphazero_amd import games
    from alphazero_amd.match import play_match

    cls = {"tictactoe": games.TicTacToeGame, "dab": games.DotsAndBoxesGame, "gomoku9": games.Gomoku9Game}[name]
    G = 6
    out = play_match(cls, HashNet(A, 5.0), HashNet(A, 3.0), G, num_simulation=sims, record_moves=True)
    d = oracle.dims(game, n)
    shape = (d["cin"], d["H"], d["W"])
    evs = (oracle.torch_eval(HashNet(A, 5.0), shape), oracle.torch_eval(HashNet(A, 3.0), shape))
    tot = {1: 0, 0: 0, -1: 0}
    for g in range(G):
        s = np.zeros(d["numel"], np.float32)
        hist = np.zeros((8, d["numel"]), np.float32)
        player, t = 1, 0
        while True:
            a_to_move = (player == 1) == (g % 2 == 0)
            r = oracle.search(game, n, s, player, hist, sims, evs[0 if a_to_move else 1], noise=None)
            a = int(np.argmax(r["visits"]))  # first maximum, like torch.argmax on CPU
            assert int(out["moves"][t][g]) == a, (g, t)
            s, nxt = oracle.next_state(game, n, s, a, player)
            hist = np.concatenate([hist[1:], s[None]])
            t += 1
            over, winner = oracle.status(game, n, s)[:2]
            if over:
                a_colour = 1 if g % 2 == 0 else -1
                want = 0 if winner == 0 else (1 if winner == a_colour else -1)
                assert int(out["results"][g]) == want and int(out["plies"][g]) == t
                tot[want] += 1
                break
            player = nxt
        assert all(int(m[g]) == -1 for m in out["moves"][t:])
    assert (out["a_wins"], out["draws"], out["b_wins"]) == (tot[1], tot[0], tot[-1])


@pytest.mark.parametrize("name", ["dab", "gomoku9-quick"])
def test_match_on_the_fused_f32_networks_replays_like_the_oracle(oracle, name):
    """play_match(fused=True): both networks on the hand-written f32 kernels, each evaluating the rows flagged for it from the
    packed leaf batch; every game replayed on the oracle with the same kernels as its evaluators, move for move."""
    from alphazero_amd import games, nets
    from alphazero_amd.fused import FusedTowerEvaluator, SmallNetEvaluator
    from alphazero_amd.match import play_match

    from .test_gpu_bench_path import _tower_callback
    from .test_gpu_tower import _randomised

    if name == "dab":
        game, n, sims, cls = 1, 3, 20, games.DotsAndBoxesGame
        torch.manual_seed(1)
        ma = nets.DotsAndBoxesNet().eval().cuda()
        torch.manual_seed(2)
        mb = nets.DotsAndBoxesNet().eval().cuda()
        make = SmallNetEvaluator
    else:
        game, n, sims, cls = 2, 9, 14, games.Gomoku9Game
        torch.manual_seed(1)
        ma = _randomised(nets.QuickNet(9, 9, 64, 1, 2)).eval().cuda()
        torch.manual_seed(2)
        mb = _randomised(nets.GomokuNet(9, 64, 1)).eval().cuda()
        make = lambda m: FusedTowerEvaluator(m, "f32")
    G = 5
    out = play_match(cls, ma, mb, G, num_simulation=sims, record_moves=True, fused=True)
    d = oracle.dims(game, n)
    shape = (d["cin"], d["H"], d["W"])
    evs = (_tower_callback(make(ma), shape, torch.float32)[0], _tower_callback(make(mb), shape, torch.float32)[0])
    for g in range(G):
        s = np.zeros(d["numel"], np.float32)
        hist = np.zeros((8, d["numel"]), np.float32)
        player, t = 1, 0
        while True:
            a_to_move = (player == 1) == (g % 2 == 0)
            r = oracle.search(game, n, s, player, hist, sims, evs[0 if a_to_move else 1], noise=None)
            a = int(np.argmax(r["visits"]))
            assert int(out["moves"][t][g]) == a, (g, t)
            s, nxt = oracle.next_state(game, n, s, a, player)
            hist = np.concatenate([hist[1:], s[None]])
            t += 1
            over, winner = oracle.status(game, n, s)[:2]
            if over:
                a_colour = 1 if g % 2 == 0 else -1
                want = 0 if winner == 0 else (1 if winner == a_colour else -1)
                assert int(out["results"][g]) == want and int(out["plies"][g]) == t
                break
            player = nxt
