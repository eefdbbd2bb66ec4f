"""The reference's scripted rule cases (tests/golden/rules_*.npz, recorded from the reference's own game classes:
fives in all four directions, board-edge and anti-diagonal lines, six in a row, second-player wins, dots-and-boxes
double completions and extra turns, full-board draws) replayed through the HIP rule kernels: every game of a fixture
runs in its own slot, its moves are forced through az_play_moves, and after every ply the position (newest plane of
az_root_nn_state), the player to move, the terminal flag, the result and the legal-move set (support of the root prior
of the next search) are compared with the recording, bit for bit.  examples/gomoku/game.py:62-89,
examples/dots_and_boxes/game.py:126-156, examples/tictactoe/game.py:20-77."""
import os

import numpy as np
import pytest
import torch

from .conftest import DAB_RULES, GOLDEN, MNK_RULES, cols_of, win_of

pytestmark = pytest.mark.gpu

RULES = [("tictactoe", 0, 3), ("dab", 1, 3), ("gomoku15", 2, 15), ("gomoku9", 2, 9)]
# + the full-board draw (81 plies, no five for either side: terminal, no winner, result 0; examples/gomoku/game.py:47-52)
# + gomoku under other values of the reference's BOARD_SIZE / WIN_LENGTH constants (examples/gomoku/game.py:21-22)
# + dots-and-boxes under other values of BOX_ROWS / BOX_COLS (examples/dots_and_boxes/game.py:24-25)
FIXTURES = RULES + [("gomoku9_draw", 2, 9)] + MNK_RULES + DAB_RULES


def _two_sims(eng, nn, logits, values):
    """expand the root (simulation 1) and visit one child (simulation 2): az_play_moves needs a searched root"""
    eng.begin_search()
    for _ in range(2):
        eng.select_gather(nn)
        eng.eval_hash(nn, logits, values, 4.0)
        eng.expand_backup(logits, values, None)


@pytest.mark.parametrize("name,game,n", FIXTURES)
def test_reference_rule_cases_through_the_hip_rules(name, game, n):
    from alphazero_amd import Engine

    z = np.load(os.path.join(GOLDEN, f"rules_{name}.npz"))
    G = int(z["n_games"])
    eng = Engine(game, G, 4, board_size=n, device="cuda:0", win_length=win_of(name), board_cols=cols_of(name))
    d = eng.dims
    A, numel = d.n_action, d.state_numel
    nn, logits, values = eng.new_nn_input(), eng.new(G, A), eng.new(G)
    acts = [z[f"g{g}_actions"] for g in range(G)]
    it = torch.ones(G, device="cuda")
    expo = torch.ones(G, A, device="cuda")
    prior = eng.new(G, A)
    alive = np.ones(G, bool)  # scripted and not yet over
    for k in range(max(len(a) for a in acts)):
        _two_sims(eng, nn, logits, values)
        eng.root_policy(root_prior=prior)
        pr = prior.cpu().numpy()
        for g in range(G):
            if alive[g] and k < len(acts[g]):
                mask = z[f"g{g}_mask0"] if k == 0 else z[f"g{g}_masks"][k - 1]
                assert np.array_equal(pr[g] > 0, mask > 0), (name, g, k, "legal moves")
        forced = torch.tensor([int(acts[g][k]) if (alive[g] and k < len(acts[g])) else -1 for g in range(G)], dtype=torch.int32,
                              device="cuda")
        _, movers, _, _ = eng.game_info()
        movers = movers.cpu().numpy()
        a, done = eng.play_moves(it, expo, forced)
        a, done = a.cpu().numpy(), done.cpu().numpy()
        # a finished slot is inactive: its newest plane is read before anything resets it
        _, player, active, result = eng.game_info()
        player, active, result = player.cpu().numpy(), active.cpu().numpy(), result.cpu().numpy()
        st = eng.root_nn_state().cpu().numpy().reshape(G, d.nn_channels, -1)
        for g in range(G):
            if not (alive[g] and k < len(acts[g])):
                continue
            assert a[g] == acts[g][k]
            want_term = bool(z[f"g{g}_terminal"][k])
            assert bool(done[g]) == want_term and bool(active[g]) == (not want_term), (name, g, k, "terminal")
            assert player[g] == z[f"g{g}_next_players"][k], (name, g, k, "next player")
            if not want_term:  # (an inactive slot's planes are not refreshed)
                newest = st[g, 7 * d.state_channels:8 * d.state_channels].reshape(-1)
                assert np.array_equal(newest.astype(np.int8), z[f"g{g}_states"][k].reshape(-1)), (name, g, k, "state")
                assert st[g, -1, 0] == player[g]
            else:
                w = int(z[f"g{g}_winner"][k])  # +1 / -1 / 0 (no winner); result is from the last mover's view
                assert result[g] == (0.0 if w == 0 else (1.0 if w == movers[g] else -1.0)), (name, g, k, "result")
                alive[g] = False
    eng.check()
    if name.endswith("_draw"):
        assert not alive[0] and len(acts[0]) == A  # the board filled up and the game ended as a draw on the last ply
    # the final position of every finished game, from its trajectory: emitted sample 0 is the last position PLAYED FROM,
    # so replay the terminal board through the sample path instead: states of all plies via az_emit_samples
    fin = [g for g in range(G) if bool(z[f"g{g}_terminal"][-1])]
    if fin:
        plies = [len(acts[g]) for g in fin]
        states, pis, zs, offs = eng.emit_samples(fin, plies)
        states = states.cpu().numpy().reshape(-1, d.nn_channels, numel // d.state_channels)
        for i, g in enumerate(fin):
            lo = int(offs[i])
            for j in range(plies[i]):  # sample j is ply t = T-1-j: its newest plane is the position BEFORE move t
                t = plies[i] - 1 - j
                want = z[f"g{g}_states"][t - 1].reshape(-1) if t > 0 else np.zeros(numel, np.int8)
                got = states[lo + j, 7 * d.state_channels:8 * d.state_channels].reshape(-1)
                assert np.array_equal(got.astype(np.int8), want), (name, g, t)
        eng.check()


@pytest.mark.parametrize("name,game,n", RULES)
def test_illegal_forced_moves_raise(name, game, n):
    """occupied cell / out of range: ValueError in the reference (e.g. examples/tictactoe/game.py:31,68), the device
    error flag here"""
    from alphazero_amd import Engine
    from alphazero_amd.engine import EngineError

    for bad in ("occupied", "range"):
        eng = Engine(game, 2, 4, board_size=n, device="cuda:0")
        A = eng.dims.n_action
        nn, logits, values = eng.new_nn_input(), eng.new(2, A), eng.new(2)
        it, expo = torch.ones(2, device="cuda"), torch.ones(2, A, device="cuda")
        _two_sims(eng, nn, logits, values)
        eng.play_moves(it, expo, torch.tensor([0, 0], dtype=torch.int32, device="cuda"))
        eng.check()
        _two_sims(eng, nn, logits, values)
        eng.play_moves(it, expo, torch.tensor([0 if bad == "occupied" else A + 3, 1], dtype=torch.int32, device="cuda"))
        with pytest.raises(EngineError):
            eng.check()


def test_set_position_refuses_decided_roots():
    from alphazero_amd import Engine
    from alphazero_amd.engine import EngineError

    eng = Engine(2, 1, 4, board_size=15, device="cuda:0")
    h = np.zeros((8, 225), np.float32)
    for c in range(5):  # five in the anti-diagonal direction touching the edge
        h[7, (4 - c) * 15 + c] = -1.0
        h[7, 200 + c] = 1.0 if c < 4 else 0.0
    with pytest.raises(EngineError):
        eng.set_position(0, h, 1)
    h[7, 4 * 15] = 0.0  # four only: fine
    eng.set_position(0, h, -1)
    ttt = Engine(0, 1, 4, board_size=3, device="cuda:0")
    b = np.zeros((8, 9), np.float32)
    b[7, [0, 4, 8]] = 1.0
    b[7, [1, 2]] = -1.0
    with pytest.raises(EngineError):
        ttt.set_position(0, b, -1)
