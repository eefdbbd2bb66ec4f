"""Virtual loss / in-tree leaf batching (az_select_wave / az_expand_wave) against the oracle's restatement of the same
rule (oracle/az_oracle.c: az_oracle_search_wave; the reference carries the hooks -- core/MCTS_alphazero.py:32-33,70,
config/basic.py:19-30 -- but no code that uses them, so the rule is this repository's and is stated in both places).
K = 1 must be the sequential search bit for bit; K > 1 must match the oracle bit for bit (visits, Q, priors, pi).
PARITY UNPINNED for K > 1: no reference code, fixture or golden vector exists for this mode -- the comparison is engine vs
this repository's own CPU statement of the rule, not vs the reference; K = 1 is pinned through the sequential search."""
import numpy as np
import pytest
import torch

from .test_gpu_parity import _random_position

pytestmark = pytest.mark.gpu

CONFIGS = [("tictactoe", 0, 3, 60, 6), ("dab", 1, 3, 80, 20), ("gomoku9", 2, 9, 120, 30), ("gomoku15", 2, 15, 150, 40)]


def _run_waves(eng, sims, noise, scale, dense):
    G, A = eng.n_slots, eng.dims.n_action
    nn = eng.new_nn_input() if dense else None
    logits, values = eng.new(eng.rows, A), eng.new(eng.rows)
    eng.begin_search()
    steps, left = 0, sims
    while left > 0:
        for _ in range((left + eng.leaves_per_step - 1) // eng.leaves_per_step):
            eng.select_wave(sims, nn)
            eng.eval_hash(nn, logits, values, scale)
            eng.expand_wave(logits, values, noise)
            steps += 1
        tot, left = eng.search_remaining(sims)
        assert (tot == 0) == (left == 0)
    visits = eng.new(G, A, dtype=torch.int32)
    q = eng.new(G, A, dtype=torch.float64)
    prior = eng.new(G, A)
    pi = eng.root_policy(visits=visits, child_q=q, root_prior=prior)
    eng.check()
    return pi.cpu().numpy(), visits.cpu().numpy(), q.cpu().numpy(), prior.cpu().numpy(), steps


@pytest.mark.parametrize("name,game,n,sims,max_moves", CONFIGS)
@pytest.mark.parametrize("K,dense", [(1, True), (4, True), (8, False), (16, False)])
def test_wave_search_matches_oracle(oracle, name, game, n, sims, max_moves, K, dense):
    from alphazero_amd import Engine

    rng = np.random.default_rng(100 + K + n)
    G = 20
    d = oracle.dims(game, n)
    vl = 0.3 if K != 8 else 0.45
    eng = Engine(game, G, sims, board_size=n, device="cuda:0", nn_dtype=torch.float32 if dense else torch.bfloat16,
                 leaves_per_step=K, virtual_loss=vl)
    assert eng.leaves_per_step == K and eng.rows == G * K
    pos = [_random_position(oracle, game, n, int(rng.integers(0, max_moves + 1)), rng) for _ in range(G)]
    for g, (h, p) in enumerate(pos):
        eng.set_position(g, h, p)
    noise_np = rng.dirichlet([0.3] * d["A"], size=G).astype(np.float32)
    noise = torch.from_numpy(noise_np).cuda()
    scale = 6.0
    pi, visits, q, prior, steps = _run_waves(eng, sims, noise, scale, dense)
    fn, ctx, _keep = oracle.hash_eval(scale)
    waves = []
    for g, (h, p) in enumerate(pos):
        r = oracle.search_wave(game, n, h[-1], p, h, sims, K, fn, ctx, virtual_loss=vl, noise=noise_np[g])
        assert np.array_equal(visits[g], r["visits"]), (name, K, g)
        assert np.array_equal(q[g], r["child_q"]), (name, K, g)
        assert np.array_equal(prior[g], r["root_prior"]), (name, K, g)
        assert np.array_equal(pi[g], r["pi"]), (name, K, g)
        waves.append(int(r["stats"][4]))
        if K == 1:  # the sequential search, bit for bit
            r1 = oracle.search(game, n, h[-1], p, h, sims, fn, ctx, noise=noise_np[g])
            assert np.array_equal(r1["visits"], r["visits"]) and np.array_equal(r1["child_q"], r["child_q"])
    assert steps >= max(waves)  # the engine runs every tree until the slowest is done
    st = eng.read_stats()
    assert st["sims"] == G * sims and st["evals"] + st["terminals"] == st["sims"]
    if K > 1:
        assert np.median(waves) < sims  # leaves really are batched inside a tree (a one-move position cannot batch)


def test_sequential_entry_points_are_refused_on_a_wave_engine():
    from alphazero_amd import Engine
    from alphazero_amd.engine import EngineError

    eng = Engine(0, 4, 8, board_size=3, device="cuda:0", leaves_per_step=4)
    with pytest.raises(EngineError):
        eng.select_gather(eng.new_nn_input())


def test_runner_wave_mode_with_the_fused_tower(oracle):
    """SelfPlayRunner on a K = 4 engine: packed leaf batch of n_slots*K rows + flags -> fused tower -> expand_wave, against
    the oracle's wave search with the same tower as its evaluator (bit-exact: the tower is deterministic per board)"""
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import GomokuNet

    from .test_gpu_bench_path import _fix_noise, _results, _tower_callback
    from .test_gpu_tower import _randomised

    n, G, sims, K = 9, 24, 64, 4
    rng = np.random.default_rng(8)
    torch.manual_seed(2)
    net = _randomised(GomokuNet(n, 64, 2)).eval().cuda().bfloat16()
    eng = Engine(2, G, sims, board_size=n, nn_dtype=torch.bfloat16, device="cuda:0", leaves_per_step=K)
    pos = [_random_position(oracle, 2, n, int(rng.integers(0, 25)), rng) for _ in range(G)]
    for g, (h, p) in enumerate(pos):
        eng.set_position(g, h, p)
    run = SelfPlayRunner(eng, net, sims, fused=True)
    noise_np = rng.dirichlet([0.3] * (n * n), size=G).astype(np.float32)
    _fix_noise(run, noise_np)
    run.search()
    assert run.wave_steps < sims
    pi, visits, q, prior = _results(eng)
    d = oracle.dims(2, n)
    cb, _ = _tower_callback(run.fused, (d["cin"], d["H"], d["W"]), torch.bfloat16)
    for g, (h, p) in enumerate(pos):
        r = oracle.search_wave(2, n, h[-1], p, h, sims, K, cb, noise=noise_np[g])
        assert np.array_equal(visits[g], r["visits"]), g
        assert np.array_equal(q[g], r["child_q"]), g
