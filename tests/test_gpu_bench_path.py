"""Parity of the path bench.py actually times (BASELINE configs[3]): ``SelfPlayRunner.search()`` on gomoku 15x15 =
az_begin_search, az_select_gather(nn_input = NULL), then (evaluate, az_expand_select(nn_input = NULL)) per simulation --
the k_expand_select<Gomoku<15>, bf16> instantiation with the register hand-over between backup and next descent, the
packed leaf ring and the needs-the-network flags -- against oracle.search (core/MCTS_alphazero.py:144-178), many slots
from distinct mid-game positions.

(a) the synthetic hash evaluator reading the PACKED leaf batch: visits, Q, root priors, pi bit-exact;
(b) the fused bf16 tower on the packed + flags path, with the oracle's evaluator callback running the SAME tower on
    the single dense position: visits and pi bit-exact (the tower is deterministic per board), Q and priors bit-exact too
    since both sides then see identical logits and values;
(c) the hand-written f32 tower the same way (the headline's precision).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from .test_gpu_parity import _random_position
from .test_gpu_tower import _randomised

pytestmark = pytest.mark.gpu


def _positions(oracle, G, rng, n=15, max_moves=40):
    return [_random_position(oracle, 2, n, int(rng.integers(0, max_moves + 1)), rng) for _ in range(G)]


def _fix_noise(run, noise_np):
    t = torch.from_numpy(noise_np)

    def draw():
        run.noise.copy_(t)
        return run.noise

    run.draw_noise = draw


def _results(eng):
    G, A = eng.n_slots, eng.dims.n_action
    visits = eng.new(G, A, dtype=torch.int32)
    q = eng.new(G, A, dtype=torch.float64)
    prior = eng.new(G, A)
    pi = eng.root_policy(visits=visits, child_q=q, root_prior=prior)
    eng.check()
    return pi.cpu().numpy(), visits.cpu().numpy(), q.cpu().numpy(), prior.cpu().numpy()


@pytest.mark.parametrize("n,G,sims", [(15, 96, 128), (9, 40, 100)])
def test_runner_search_packed_hash_matches_oracle(oracle, n, G, sims):
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import GomokuNet

    rng = np.random.default_rng(21 + n)
    eng = Engine(2, G, sims, board_size=n, nn_dtype=torch.bfloat16, device="cuda:0")
    A = eng.dims.n_action
    pos = _positions(oracle, G, rng, n)
    for g, (h, p) in enumerate(pos):
        eng.set_position(g, h, p)
    run = SelfPlayRunner(eng, GomokuNet(n, 64, 1).cuda().bfloat16().eval(), sims)
    scale = 6.0
    run.evaluate = lambda: eng.eval_hash(None, run.logits, run.values, scale)  # reads the packed leaf batch
    # make run.search() take the packed route of the fused evaluator: no dense batch is ever written
    run.fused = object()
    noise_np = rng.dirichlet([0.3] * A, size=G).astype(np.float32)
    _fix_noise(run, noise_np)
    run.nn_input.fill_(7.0)
    run.search()
    assert bool((run.nn_input == 7.0).all())  # nn_input == NULL all the way
    pi, visits, q, prior = _results(eng)
    fn, ctx, _keep = oracle.hash_eval(scale)
    for g, (h, p) in enumerate(pos):
        r = oracle.search(2, n, h[-1], p, h, sims, fn, ctx, noise=noise_np[g])
        assert np.array_equal(visits[g], r["visits"]), g
        assert np.array_equal(q[g], r["child_q"]), g
        assert np.array_equal(prior[g], r["root_prior"]), g
        assert np.array_equal(pi[g], r["pi"]), g
    st = eng.read_stats()
    assert st["sims"] == G * sims and st["evals"] + st["terminals"] == st["sims"]


def _tower_callback(ev, shape, dtype):
    """oracle evaluator: the fused tower on ONE dense position (batch 1, the dense-input route of the same kernel)"""
    from oracle import az_oracle as O

    cin, h, w = shape
    calls = [0]

    def _cb(_ctx, x, n_in, logits, n_action, value):
        arr = np.ctypeslib.as_array(x, shape=(n_in,))
        xt = torch.from_numpy(arr.copy()).view(1, cin, h, w).cuda().to(getattr(ev, "in_dtype", dtype)).contiguous()  # (the tower's own dense dtype)
        p, v = ev(xt)
        np.ctypeslib.as_array(logits, shape=(n_action,))[:] = p.reshape(-1).cpu().numpy()
        value[0] = float(v.reshape(-1)[0].item())
        calls[0] += 1

    return O.EVAL_FN(_cb), calls


@pytest.mark.parametrize("precision", ["bf16", "f32", "f32w", "f32-quick", "bf16-quick", "f16x2", "bf16r", "f16"])
def test_runner_search_fused_tower_matches_oracle(oracle, precision):
    """engine + tower on the packed/flags path == oracle + the same tower as its evaluator ("f32-quick": the reference's
    QuickModel family, 128 channels, on the f32 tower's QuickModel instantiation -- config 5's network; "f16x2": the
    split-operand tower of az_tower_h2.hip on an f32 engine; "bf16r" / "f16": the row-tile tower of az_tower_h1.hip, likewise;
    "f32w": the Winograd f32 tower of az_tower_w.hip -- what fused=True picks for this network on an f32 engine, the bench's primary line;
    "f32": the direct-convolution f32 tower by name)"""
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import GomokuNet, QuickNet

    n, G, sims = 15, 64, 100
    rng = np.random.default_rng(5)
    torch.manual_seed(3)
    dtype = torch.bfloat16 if precision in ("bf16", "bf16-quick") else torch.float32  # (the towers asked for by name: f32 engine)
    net = _randomised(QuickNet(n, 9, 128, 2, 3) if precision.endswith("quick") else GomokuNet(n, 64, 2)).eval().cuda().to(dtype)
    eng = Engine(2, G, sims, board_size=n, nn_dtype=dtype, device="cuda:0")
    A = eng.dims.n_action
    pos = _positions(oracle, G, rng, n)
    for g, (h, p) in enumerate(pos):
        eng.set_position(g, h, p)
    # ("bf16" by name: az_tower.hip -- left to itself the runner picks the row-tile tower for this network on a bf16 engine)
    run = SelfPlayRunner(eng, net, sims, fused=precision if precision in ("f16x2", "bf16r", "f16", "bf16", "f32") else True)
    assert run.fused is not None and run.fused.precision == precision.split("-")[0]
    noise_np = rng.dirichlet([0.3] * A, size=G).astype(np.float32)
    _fix_noise(run, noise_np)
    run.search()
    pi, visits, q, prior = _results(eng)
    d = oracle.dims(2, n)
    cb, calls = _tower_callback(run.fused, (d["cin"], d["H"], d["W"]), dtype)
    for g, (h, p) in enumerate(pos):
        r = oracle.search(2, n, h[-1], p, h, sims, cb, noise=noise_np[g])
        assert np.array_equal(visits[g], r["visits"]), g
        assert np.array_equal(pi[g], r["pi"]), g
        assert np.array_equal(q[g], r["child_q"]), g
        assert np.array_equal(prior[g], r["root_prior"]), g
    assert calls[0] == eng.read_stats()["evals"]


def test_runner_picks_the_row_tile_tower_for_a_bf16_engine_on_15x15():
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import GomokuNet

    for n, want in ((15, "bf16r"), (9, "bf16")):
        eng = Engine(2, 8, 4, board_size=n, nn_dtype=torch.bfloat16, device="cuda:0")
        run = SelfPlayRunner(eng, GomokuNet(n, 64, 1).eval().cuda().bfloat16(), 4, fused=True)
        assert run.fused.precision == want
        run.search()
        eng.check()
        eng.close()
    eng = Engine(2, 8, 4, board_size=15, nn_dtype=torch.bfloat16, device="cuda:0")
    assert SelfPlayRunner(eng, GomokuNet(15, 128, 1).eval().cuda().bfloat16(), 4, fused=True).fused.precision == "bf16"
    eng.close()


@pytest.mark.parametrize("kind", ["gomokunet-bf16", "quicknet-f32", "quicknet-bf16"])
def test_runner_search_graph_replay_matches_eager(oracle, kind):
    """the HIP-graph replay of the simulation step (what the small configs run) == the eager loop, bit for bit"""
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import GomokuNet, QuickNet

    n, G, sims = 9, 48, 60
    rng = np.random.default_rng(9)
    torch.manual_seed(4)
    dt = torch.float32 if kind.endswith("f32") else torch.bfloat16
    net = _randomised(QuickNet(n, 9, 64, 2, 2) if kind.startswith("quick") else GomokuNet(n, 64, 2)).eval().cuda().to(dt)
    pos = _positions(oracle, G, rng, n, 20)
    noise_np = rng.dirichlet([0.3] * (n * n), size=G).astype(np.float32)
    out = []
    for graph in (False, True, 7):  # 7: graphs of seven simulation steps, 59 = 8 x 7 + 3 eager steps behind them
        eng = Engine(2, G, sims, board_size=n, nn_dtype=dt, device="cuda:0")
        for g, (h, p) in enumerate(pos):
            eng.set_position(g, h, p)
        run = SelfPlayRunner(eng, net, sims, fused=True, use_graph=bool(graph), graph_unit=None if graph is True else graph)
        _fix_noise(run, noise_np)
        run.search()
        out.append(_results(eng))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["gomokunet-f32", "quicknet-bf16", "dab-smallnet", "gomokunet15-rowtile-bf16", "gomokunet15-rowtile-f16",
                                  "gomokunet15-winograd-f32"])
def test_graph_replay_after_weight_refresh_matches_eager(oracle, kind):
    """A weight publish between two moves: ``refresh()`` re-folds the module's new weights into the evaluator's packed tensors.
    A search graph captured BEFORE the refresh has those tensors' addresses baked in, so the refresh must rewrite them in place
    (and re-capture only for what travels by value): the replayed search after the refresh == an eager search with a freshly
    built evaluator on the new weights, bit for bit -- and differs from the search on the old weights."""
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import DotsAndBoxesNet, GomokuNet, QuickNet

    rng = np.random.default_rng(31)
    torch.manual_seed(8)
    fused = True
    if kind == "dab-smallnet":
        game, n, G, sims, dt = 1, 3, 64, 40, torch.float32
        mk = lambda: DotsAndBoxesNet()
    elif "rowtile" in kind or "winograd" in kind:
        # az_tower_h1.hip under graph replay: its one-board-ahead hand-over, scales and selector fragments; az_tower_w.hip (what
        # fused=True picks for this network on an f32 engine): its weight stream running across board and layer boundaries
        game, n, G, sims, dt = 2, 15, 40, 30, torch.float32
        fused = True if "winograd" in kind else "bf16r" if kind.endswith("bf16") else "f16"
        mk = lambda: _randomised(GomokuNet(n, 64, 2))
    else:
        game, n, G, sims = 2, 9, 48, 40
        dt = torch.float32 if kind.endswith("f32") else torch.bfloat16
        mk = lambda: _randomised(QuickNet(n, 9, 64, 2, 2) if kind.startswith("quick") else GomokuNet(n, 64, 2))
    net = mk().eval().cuda().to(dt)
    new_weights = {k: v.clone() for k, v in mk().eval().cuda().to(dt).state_dict().items()}
    A = Engine(game, 1, 1, board_size=n, device="cuda:0").dims.n_action
    noise_np = rng.dirichlet([0.3] * A, size=G).astype(np.float32)

    def fresh_engine():
        e = Engine(game, G, sims, board_size=n, nn_dtype=dt, device="cuda:0")
        if game == 2:
            for g, (h, p) in enumerate(pos):
                e.set_position(g, h, p)
        return e

    pos = _positions(oracle, G, rng, n, 20) if game == 2 else None

    def restart(e):
        if game == 2:
            for g, (h, p) in enumerate(pos):
                e.set_position(g, h, p)
        else:
            e.reset_games(None)

    eng = fresh_engine()
    run = SelfPlayRunner(eng, net, sims, fused=fused, use_graph=True)
    assert "winograd" not in kind or run.fused.precision == "f32w"
    _fix_noise(run, noise_np)
    run.search()  # captures the graph on the OLD weights
    old = _results(eng)
    g0, gen0 = run._graph, run.fused.generation
    assert g0 is not None
    names = ("w1", "w2", "head", "w0", "wl", "bias", "headw", "wp", "w1p", "w2p", "sw")
    ptrs = {k: getattr(run.fused, k).data_ptr() for k in names if isinstance(getattr(run.fused, k, None), torch.Tensor)}
    assert ptrs
    net.load_state_dict(new_weights)
    run.fused.refresh()
    assert all(getattr(run.fused, k).data_ptr() == p for k, p in ptrs.items()), "refresh() must rewrite the packed weights in place"
    restart(eng)
    run.search()  # same runner, same engine: the old graph replays unless a by-value argument changed (then it is re-captured)
    assert (run._graph is g0) == (run.fused.generation == gen0)
    replayed = _results(eng)
    eng.close()

    eng = fresh_engine()
    run2 = SelfPlayRunner(eng, net, sims, fused=fused, use_graph=False)
    _fix_noise(run2, noise_np)
    run2.search()
    eager = _results(eng)
    eng.close()
    for a, b in zip(replayed, eager):
        assert np.array_equal(a, b)
    assert not np.array_equal(old[1], eager[1]), "the two weight sets must lead to different searches for the test to mean anything"


@pytest.mark.parametrize("precision", ["f32", "bf16", "f32-quick", "bf16-quick"])
def test_full_size_bench_path_properties(oracle, precision):
    """BASELINE configs[3] at its full size -- 16,384 gomoku-15 trees through SelfPlayRunner.search() with the fused tower on
    the packed + flags path -- checked through size-independent properties: two distinct root positions alternate over the
    slots, so (no noise, deterministic tower) every even slot must grow the same tree and every odd slot the same tree,
    whichever workgroup, ticket or launch evaluated its leaves; both trees equal the oracle's with the same tower as its
    evaluator; visit sums, pi normalisation and the engine's counters hold for all slots."""
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner
    from alphazero_amd.nets import GomokuNet, QuickNet

    n, G, sims = 15, 16384, 20
    rng = np.random.default_rng(77)
    torch.manual_seed(5)
    dtype = torch.bfloat16 if precision.startswith("bf16") else torch.float32
    # (the QuickModel towers: every workgroup of a full grid runs dozens of boards through its stash region / residual registers)
    net = _randomised(QuickNet(n, 9, 64, 1, 3) if precision.endswith("quick") else GomokuNet(n, 64, 2)).eval().cuda().to(dtype)
    eng = Engine(2, G, sims, board_size=n, nn_dtype=dtype, device="cuda:0")
    A = eng.dims.n_action
    pos = [_random_position(oracle, 2, n, m, rng) for m in (9, 30)]
    for g in range(G):
        eng.set_position(g, *pos[g & 1])
    run = SelfPlayRunner(eng, net, sims, fused=True, use_dirichlet=False)
    run.search()
    pi, visits, q, prior = _results(eng)
    assert (visits.sum(1) == sims - 1).all()
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5)
    for par in (0, 1):
        assert (visits[par::2] == visits[par]).all(), par
        assert (q[par::2] == q[par]).all() and (prior[par::2] == prior[par]).all(), par
    st = eng.read_stats()
    assert st["sims"] == G * sims and st["evals"] + st["terminals"] == st["sims"]
    d = oracle.dims(2, n)
    cb, _calls = _tower_callback(run.fused, (d["cin"], d["H"], d["W"]), dtype)
    for par, (h, p) in enumerate(pos):
        r = oracle.search(2, n, h[-1], p, h, sims, cb)
        assert np.array_equal(visits[par], r["visits"]) and np.array_equal(q[par], r["child_q"])
        assert np.array_equal(prior[par], r["root_prior"]) and np.array_equal(pi[par], r["pi"])
