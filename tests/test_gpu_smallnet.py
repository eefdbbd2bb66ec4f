"""csrc/az_smallnet.hip -- the reference's tic-tac-toe / dots-and-boxes networks (examples/tictactoe/model.py,
examples/dots_and_boxes/model.py) as one f32 MFMA kernel per evaluation -- through the C ABI (az_smallnet_forward_f32):

(a) against the module's own PyTorch float32 forward on the same inputs (tolerance 2e-5 absolute: f32 operands and
    accumulation on both sides, only the summation order differs);
(b) the packed-leaf-batch + flags route of the engine == the dense route, bit for bit, and unflagged rows untouched;
(c) SelfPlayRunner.search() with the kernel as evaluator == oracle.search (core/MCTS_alphazero.py:144-178) with the same
    kernel as the oracle's evaluator callback: visits, Q, priors, pi bit-exact over many slots.
"""
import numpy as np
import pytest
import torch

from .test_gpu_bench_path import _fix_noise, _results, _tower_callback
from .test_gpu_parity import _random_position

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _net(game, seed=0):
    from alphazero_amd.nets import DotsAndBoxesNet, TicTacToeNet

    torch.manual_seed(seed)
    net = (TicTacToeNet, DotsAndBoxesNet)[game]()
    with torch.no_grad():  # default init leaves the biases tiny: make every term of the network count
        for p in net.parameters():
            if p.dim() == 1:
                p.uniform_(-0.3, 0.3)
    return net.eval().cuda()


@pytest.mark.parametrize("game,batch", [(0, 1), (0, 7), (0, 8), (0, 1000), (1, 1), (1, 3), (1, 4), (1, 3001)])
def test_smallnet_matches_fp32_forward(game, batch):
    from alphazero_amd.fused import SmallNetEvaluator

    net = _net(game)
    assert SmallNetEvaluator.supports(net)
    ev = SmallNetEvaluator(net)
    cin, hw = ((9, 3), (25, 4))[game]
    g = torch.Generator(device="cuda").manual_seed(batch)
    x = torch.randint(-1, 2, (batch, cin, hw, hw), device="cuda", generator=g).float()
    p, v = ev(x)
    p, v = p.clone(), v.clone()  # the evaluator's own output buffers are reused by the next call
    with torch.no_grad():
        pr, vr = net(x)
    assert p.shape == pr.shape and v.shape == vr.shape
    assert float((p - pr).abs().max()) < TOL * max(1.0, float(pr.abs().max()))
    assert float((v - vr).abs().max()) < TOL
    assert float(pr.abs().max()) > 0.05 and float(vr.abs().max()) > 0.01  # not a comparison of zeros
    # a board's result does not depend on where in the batch (which tile, which column) it sits
    perm = torch.randperm(batch, device="cuda", generator=g)
    p2, v2 = ev(x[perm].contiguous(), logits_out=torch.empty_like(p))
    assert torch.equal(p2, p[perm]) and torch.equal(v2.clone(), v[perm])


def test_smallnet_rejects_other_models_and_bad_input():
    from alphazero_amd.engine import EngineError
    from alphazero_amd.fused import SmallNetEvaluator
    from alphazero_amd.nets import GomokuNet

    assert not SmallNetEvaluator.supports(GomokuNet(9, 64, 1))
    with pytest.raises(EngineError):
        SmallNetEvaluator(GomokuNet(9, 64, 1).cuda())
    ev = SmallNetEvaluator(_net(0))
    with pytest.raises(EngineError):
        ev(torch.zeros(4, 9, 3, 3, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(EngineError):
        ev(torch.zeros(4, 25, 4, 4, device="cuda"))
    with pytest.raises(EngineError):
        ev(None)


@pytest.mark.parametrize("game", [0, 1])
def test_smallnet_packed_equals_dense(oracle, game):
    from alphazero_amd import Engine
    from alphazero_amd.fused import SmallNetEvaluator

    G = 203
    rng = np.random.default_rng(3 + game)
    eng = Engine(game, G, 8, nn_dtype=torch.float32, device="cuda:0")
    A = eng.dims.n_action
    max_moves = (6, 22)[game]
    for g in range(G):
        h, p = _random_position(oracle, game, 0, int(rng.integers(0, max_moves + 1)), rng)
        eng.set_position(g, h, p)
    ev = SmallNetEvaluator(_net(game, 1))
    nn = eng.new_nn_input()
    logits, values = eng.new(G, A), eng.new(G)
    ring, leaf, _, _, flags = eng.leaf_buffers()
    eng.begin_search()
    before = eng.read_stats()["evals"]
    eng.select_gather(nn)
    seen_terminal = False
    for step in range(7):  # deeper leaves each step, some of them terminal (flag 0)
        after = eng.read_stats()["evals"]
        n_eval, before = after - before, after
        pd, vd = ev(nn)
        pd, vd = pd.clone(), vd.clone().view(-1)
        pp = torch.full((G, A), 123.0, device="cuda")
        ev._buf[G][1].fill_(9.0)
        _, vp = ev(None, logits_out=pp, packed=(ring, leaf, None, None, flags), n_boards=G)
        vp = vp.view(-1)
        on = pp[:, 0] != 123.0
        assert int(on.sum()) == n_eval and 0 < n_eval <= G
        seen_terminal |= n_eval < G
        assert torch.equal(on, vp != 9.0)
        assert torch.equal(pp[on], pd[on]) and torch.equal(vp[on], vd[on])
        assert bool((pp[~on] == 123.0).all())
        logits.copy_(pd)
        values.copy_(vd)
        eng.expand_select(logits, values, None, nn)
    eng.check()
    assert seen_terminal


@pytest.mark.parametrize("game,G,sims", [(0, 70, 50), (1, 66, 100)])
def test_runner_search_smallnet_matches_oracle(oracle, game, G, sims):
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner

    rng = np.random.default_rng(11 + game)
    net = _net(game, 2)
    eng = Engine(game, G, sims, nn_dtype=torch.float32, device="cuda:0")
    A = eng.dims.n_action
    pos = [_random_position(oracle, game, 0, int(rng.integers(0, (5, 14)[game] + 1)), rng) for _ in range(G)]
    for g, (h, p) in enumerate(pos):
        eng.set_position(g, h, p)
    run = SelfPlayRunner(eng, net, sims, fused=True)
    assert type(run.fused).__name__ == "SmallNetEvaluator"
    noise_np = rng.dirichlet([0.3] * A, size=G).astype(np.float32)
    _fix_noise(run, noise_np)
    run.search()
    pi, visits, q, prior = _results(eng)
    d = oracle.dims(game, 0)
    cb, calls = _tower_callback(run.fused, (d["cin"], d["H"], d["W"]), torch.float32)
    for g, (h, p) in enumerate(pos):
        r = oracle.search(game, 0, h[-1], p, h, sims, cb, noise=noise_np[g])
        assert np.array_equal(visits[g], r["visits"]), g
        assert np.array_equal(pi[g], r["pi"]), g
        assert np.array_equal(q[g], r["child_q"]), g
        assert np.array_equal(prior[g], r["root_prior"]), g
    assert calls[0] == eng.read_stats()["evals"]


def test_smallnet_graph_replay_matches_eager(oracle):
    """BASELINE configs 1-2 run the simulation step as a HIP graph: replay == eager, bit for bit"""
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner

    game, G, sims = 1, 90, 50
    rng = np.random.default_rng(2)
    net = _net(game, 3)
    pos = [_random_position(oracle, game, 0, int(rng.integers(0, 12)), rng) for _ in range(G)]
    noise_np = rng.dirichlet([0.3] * 24, size=G).astype(np.float32)
    out = []
    for graph in (False, True):
        eng = Engine(game, G, sims, nn_dtype=torch.float32, device="cuda:0")
        for g, (h, p) in enumerate(pos):
            eng.set_position(g, h, p)
        run = SelfPlayRunner(eng, net, sims, fused=True, use_graph=graph)
        _fix_noise(run, noise_np)
        run.search()
        out.append(_results(eng))
    for a, b in zip(*out):
        assert np.array_equal(a, b)
