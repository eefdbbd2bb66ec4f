#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Only runs in the build container, where /root/reference exists (it is read-only and never travels
to the GPU box); the .npz files it writes are plain data (inputs + expected outputs) and are what
the tests read.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

How the reference is imported:
  * `colorlog` is not installed -> a 10-line stand-in module is put in sys.modules first;
  * MCTSNode._next_state (core/MCTS_alphazero.py:99) calls next_state with 2 arguments although every
    game takes 3 -> patched at call time to pass the node's own player (the reference is not edited);
  * RNG draws are captured as data: every Dirichlet sample (core/MCTS_alphazero.py:187) is recorded,
    and every torch.multinomial call (training/self_play.py:74) is replayed as
    argmax(weights / Exponential(1) draws) from the same generator state, asserted equal to torch's
    own answer, and the exponential draws recorded;
  * every model forward is recorded as a table  nn_input bytes -> (logits, value).
"""
from __future__ import annotations

import logging
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("AZ_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    stub = types.ModuleType("colorlog")

    class ColoredFormatter(logging.Formatter):
        def __init__(self, fmt=None, datefmt=None, style="%", log_colors=None, reset=True,
                     secondary_log_colors=None, **kw):
            super().__init__(None, datefmt)

    stub.ColoredFormatter = ColoredFormatter
    stub.StreamHandler = logging.StreamHandler
    sys.modules["colorlog"] = stub
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    logging.disable(logging.CRITICAL)
    import core.MCTS_alphazero as M  # noqa

    M.MCTSNode._next_state = lambda self, a: self.game_cls.next_state(self.state, a, self.player)
    return M


M = _import_reference()
from examples.tictactoe.game import TicTacToeGame  # noqa: E402
from examples.tictactoe.model import TicTacToeModel  # noqa: E402
from examples.dots_and_boxes.game import DotsAndBoxesGame  # noqa: E402
from examples.dots_and_boxes.model import DotsAndBoxesModel  # noqa: E402
from examples.gomoku.game import GomokuGame  # noqa: E402
from examples.gomoku.model import GomokuModel  # noqa: E402
from nn_models.base import BaseModel  # noqa: E402
from training.self_play import ChessArena  # noqa: E402
import training.self_play as SP  # noqa: E402

torch.set_num_threads(1)


# ----------------------------------------------------------------------------- recorders ----
class Recorder:
    def __init__(self):
        self.reset()

    def reset(self):
        self.noise = []
        self.expo = []
        self.mult_w = []
        self.mult_a = []
        self.eval_in = []
        self.eval_logits = []
        self.eval_value = []
        self.sel_trace = []  # per search: list per sim of chosen actions
        self.priors = []  # every _get_prior output, in call order
        self._cur = None


REC = Recorder()

_RealDirichlet = M.Dirichlet


class RecDirichlet(_RealDirichlet):
    def sample(self, *a, **k):
        s = super().sample(*a, **k)
        REC.noise.append(s.reshape(-1).numpy().copy())
        return s


M.Dirichlet = RecDirichlet

_real_multinomial = torch.multinomial


def rec_multinomial(w, num_samples=1, *a, **k):
    st = torch.get_rng_state()
    e = torch.empty_like(w).exponential_(1)
    mine = int(torch.argmax(w / e, dim=-1).reshape(-1)[0])
    torch.set_rng_state(st)
    out = _real_multinomial(w, num_samples, *a, **k)
    assert int(out.reshape(-1)[0]) == mine, "multinomial != argmax(w/expo)"
    REC.expo.append(e.reshape(-1).numpy().copy())
    REC.mult_w.append(w.reshape(-1).numpy().copy())
    REC.mult_a.append(mine)
    return out


torch.multinomial = rec_multinomial

_real_select = M.MCTSNode.select


def rec_select(self, c_puct=1.0):
    ch = _real_select(self, c_puct)
    act = next(a for a, c in self.children.items() if c is ch)
    REC._cur.append(act)
    return ch


M.MCTSNode.select = rec_select

_real_get_prior = M.MCTS._get_prior


def rec_get_prior(self, policy, mask, use):
    p = _real_get_prior(self, policy, mask, use)
    REC.priors.append(p.reshape(-1).numpy().copy())
    return p


M.MCTS._get_prior = rec_get_prior

_real_backward = M.MCTS._bacaward


def rec_backward(self, node, value):
    REC.sel_trace.append(list(REC._cur))
    REC._cur = []
    return _real_backward(self, node, value)


M.MCTS._bacaward = rec_backward


class RecModel(torch.nn.Module):
    """Wraps a reference model; logs every forward."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, x):
        p, v = self.inner(x)
        REC.eval_in.append(x.reshape(-1).numpy().astype(np.int8).copy())
        REC.eval_logits.append(p.reshape(-1).numpy().copy())
        REC.eval_value.append(np.float32(v.reshape(-1)[0].item()))
        return p, v


class Gomoku9Model(BaseModel):
    """The reference has no 9x9 net (GomokuModel hard-codes 15); a small net of the same contract."""

    def __init__(self):
        super().__init__()
        self._s = torch.nn.Sequential(torch.nn.Conv2d(9, 8, 3, padding=1), torch.nn.ReLU())
        self._p = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(8 * 81, 81))
        self._v = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(8 * 81, 1), torch.nn.Tanh())

    @property
    def shared_layers(self):
        return self._s

    def policy_head(self, x):
        return self._p(x)

    def value_head(self, x):
        return self._v(x)


# ----------------------------------------------------------------------------- rules KATs ----
def playout(game_cls, rng, script=None, max_moves=400):
    """Random (or scripted) playout through the STATIC interface; returns per-step arrays."""
    s = game_cls.initial_state()
    p = 1
    acts, states, nps, masks, terms, winners = [], [], [], [], [], []
    masks0 = game_cls.legal_action_mask(s).reshape(-1).numpy().copy()
    k = 0
    while not game_cls.is_terminal(s) and k < max_moves:
        m = game_cls.legal_action_mask(s).reshape(-1).numpy()
        legal = np.nonzero(m)[0]
        a = int(script[k]) if script is not None and k < len(script) else int(rng.choice(legal))
        s, p2 = game_cls.next_state(s, a, p)
        acts.append(a)
        states.append(s.reshape(-1).numpy().astype(np.int8).copy())
        nps.append(p2)
        masks.append(game_cls.legal_action_mask(s).reshape(-1).numpy().copy())
        terms.append(bool(game_cls.is_terminal(s)))
        w = game_cls._get_winner(s)
        winners.append(0 if w is None else int(w))
        p = p2
        k += 1
        if script is not None and k >= len(script):
            break
    return dict(actions=np.array(acts, np.int32), states=np.array(states, np.int8),
                next_players=np.array(nps, np.int32), masks=np.array(masks, np.float32),
                terminal=np.array(terms, np.bool_), winner=np.array(winners, np.int32), mask0=masks0)


def rules_fixture(name, game_cls, scripts, n_random, seed):
    rng = np.random.default_rng(seed)
    out = {}
    idx = 0
    for sc in scripts:
        for k, v in playout(game_cls, rng, script=sc).items():
            out[f"g{idx}_{k}"] = v
        idx += 1
    for _ in range(n_random):
        for k, v in playout(game_cls, rng).items():
            out[f"g{idx}_{k}"] = v
        idx += 1
    out["n_games"] = np.int32(idx)
    # illegal-move behaviour: occupied cell / out of range raise ValueError
    s = game_cls.initial_state()
    s, _ = game_cls.next_state(s, 0, 1)
    for a, key in ((0, "err_occupied"), (-1, "err_neg"), (10 ** 6, "err_big")):
        try:
            game_cls.next_state(s, a, -1)
            out[key] = np.bool_(False)
        except ValueError:
            out[key] = np.bool_(True)
    np.savez_compressed(os.path.join(OUT, f"rules_{name}.npz"), **out)
    print("rules", name, idx, "games")


def gomoku_scripts(n):
    def rc(r, c):
        return r * n + c

    S = []
    # horizontal five by P1 on row 0 (P2 plays row n-1)
    S.append([x for i in range(5) for x in (rc(0, i), rc(n - 1, i))][:9])
    # vertical five on last column
    S.append([x for i in range(5) for x in (rc(i, n - 1), rc(i, 0))][:9])
    # main diagonal five ending in the bottom-right corner
    S.append([x for i in range(5) for x in (rc(n - 5 + i, n - 5 + i), rc(0, i))][:9])
    # anti-diagonal five touching the left edge
    S.append([x for i in range(5) for x in (rc(i, 4 - i), rc(n - 1, i))][:9])
    # six in a row: P1 plays 0,1,2,4,5 then 3  (P2 elsewhere, no five)
    p1 = [rc(2, 0), rc(2, 1), rc(2, 2), rc(2, 4), rc(2, 5), rc(2, 3)]
    p2 = [rc(n - 1, 0), rc(n - 2, 2), rc(n - 1, 4), rc(n - 2, 6), rc(n - 1, 8)]
    S.append([x for i in range(6) for x in ((p1[i], p2[i]) if i < 5 else (p1[i],))])
    # P2 wins (second player) with a vertical five
    S.append([x for i in range(5) for x in (rc(n - 1, 2 * i if 2 * i < n else i), rc(i + 1, 3))][:10])
    return S


def make_rules():
    ttt_scripts = [[0, 3, 1, 4, 2], [0, 1, 3, 4, 8, 7], [0, 1, 2, 4, 3, 5, 7, 6, 8], [4, 0, 2, 6, 3, 5, 1, 7, 8]]
    rules_fixture("tictactoe", TicTacToeGame, ttt_scripts, 40, 1)
    # DAB: a chain that completes two boxes with one edge (double completion, extra turns)
    # H edges: a=r*3+c (r 0..3, c 0..2); V edges: a=12+r*4+c (r 0..2, c 0..3).
    # boxes (0,0) and (0,1) share V(0,1)=13: lay all their other edges first, then 13 completes both.
    dab_double = [0, 1, 3, 4, 12, 14, 13]
    dab_scripts = [dab_double, list(range(24)), list(range(23, -1, -1))]
    rules_fixture("dab", DotsAndBoxesGame, dab_scripts, 40, 2)
    GomokuGame.BOARD_SIZE = 15
    rules_fixture("gomoku15", GomokuGame, gomoku_scripts(15), 6, 3)
    GomokuGame.BOARD_SIZE = 9
    rules_fixture("gomoku9", GomokuGame, gomoku_scripts(9), 10, 4)
    GomokuGame.BOARD_SIZE = 15


# ----------------------------------------------------------------------------- search KATs ----
def random_position(game_cls, n_moves, rng):
    g = game_cls()
    for _ in range(n_moves):
        m = g.get_legal_mask().reshape(-1).numpy()
        if g.step(int(rng.choice(np.nonzero(m)[0]))):
            return None
    return g


def search_case(game_cls, model, n_moves, num_sim, use_dir, rng, seed):
    while True:
        g = random_position(game_cls, n_moves, rng)
        if g is not None:
            break
    REC.reset()
    REC._cur = []
    torch.manual_seed(seed)
    mcts = M.MCTS(game_cls)
    hist = g.get_history()
    root = g.get_state()
    # capture root children stats through a wrapped _bacaward: easier -> rerun a hook on the last root
    roots = []
    real_node_init = M.MCTSNode.__init__

    def node_init(self, *a, **k):
        real_node_init(self, *a, **k)
        if self.parent is None:
            roots.append(self)

    M.MCTSNode.__init__ = node_init
    try:
        pi = mcts.search(RecModel(model), root, g.get_player(), history_state=hist,
                         num_simulation=num_sim, use_Dirichlet=use_dir)
    finally:
        M.MCTSNode.__init__ = real_node_init
    rootn = roots[0]
    A = pi.shape[1]
    visits = np.zeros(A, np.int32)
    cq = np.zeros(A, np.float64)
    for a, c in rootn.children.items():
        visits[a] = c.visits
        cq[a] = c.value
    D = max(len(t) for t in REC.sel_trace) + 1
    trace = np.full((num_sim, D), -1, np.int32)
    for i, t in enumerate(REC.sel_trace):
        trace[i, 0] = len(t)
        trace[i, 1:1 + len(t)] = t
    return dict(
        root_state=root.reshape(-1).numpy().astype(np.int8), root_player=np.int32(g.get_player()),
        history=hist.reshape(-1).numpy().astype(np.int8), num_sim=np.int32(num_sim),
        noise=(np.array(REC.noise[0], np.float32) if use_dir else np.zeros(0, np.float32)),
        pi=pi.reshape(-1).numpy().copy(), visits=visits, child_q=cq,
        root_prior=rootn.prior.reshape(-1).numpy().copy(), root_value=np.float64(rootn.value),
        root_visits=np.int32(rootn.visits), trace=trace,
        priors=np.array(REC.priors, np.float32), eval_in=np.array(REC.eval_in, np.int8),
        eval_logits=np.array(REC.eval_logits, np.float32), eval_value=np.array(REC.eval_value, np.float32))


def make_search():
    rng = np.random.default_rng(7)
    cases = []

    def add(tag, game_cls, model, n_moves, num_sim, use_dir, seed):
        c = search_case(game_cls, model, n_moves, num_sim, use_dir, rng, seed)
        cases.append((tag, c))
        print("search", tag, "moves", n_moves, "sims", num_sim, "maxdepth", c["trace"][:, 0].max())

    torch.manual_seed(100)
    ttt = TicTacToeModel().eval()
    for i, (nm, ns, ud) in enumerate([(0, 50, True), (2, 50, True), (4, 80, True), (5, 120, False), (6, 50, True)]):
        add(f"tictactoe_{i}", TicTacToeGame, ttt, nm, ns, ud, 200 + i)
    torch.manual_seed(101)
    dab = DotsAndBoxesModel().eval()
    for i, (nm, ns, ud) in enumerate([(0, 80, True), (6, 80, True), (14, 80, True), (18, 200, False), (20, 80, True)]):
        add(f"dab_{i}", DotsAndBoxesGame, dab, nm, ns, ud, 300 + i)
    torch.manual_seed(102)
    gmk = GomokuModel().eval()
    GomokuGame.BOARD_SIZE = 15
    for i, (nm, ns, ud) in enumerate([(0, 60, True), (12, 60, True), (30, 40, False)]):
        add(f"gomoku15_{i}", GomokuGame, gmk, nm, ns, ud, 400 + i)
    torch.manual_seed(103)
    g9 = Gomoku9Model().eval()
    GomokuGame.BOARD_SIZE = 9
    for i, (nm, ns, ud) in enumerate([(0, 100, True), (10, 200, True), (40, 100, False)]):
        add(f"gomoku9_{i}", GomokuGame, g9, nm, ns, ud, 500 + i)
    GomokuGame.BOARD_SIZE = 15
    out = {"tags": np.array([t for t, _ in cases])}
    for t, c in cases:
        for k, v in c.items():
            out[f"{t}__{k}"] = v
    np.savez_compressed(os.path.join(OUT, "search.npz"), **out)


# --------------------------------------------------------------------------- self-play KATs ----
class ListPool:
    def __init__(self):
        self.items = []

    def put_tupule_experience(self, exp):
        self.items.append(exp)


def selfplay_case(game_cls, model_cls, num_sim, tao, seed, model_seed):
    import config

    old = M.MCTS.search.__defaults__
    # TRAIN_NUM_SIMULATION is bound as a default argument (core/MCTS_alphazero.py:134)
    M.MCTS.search.__defaults__ = (old[0], num_sim) + tuple(old[2:])
    try:
        torch.manual_seed(model_seed)
        arena = ChessArena(model_cls, game_cls, torch.device("cpu"))
        arena.model = RecModel(arena.model)
        REC.reset()
        REC._cur = []
        torch.manual_seed(seed)
        pool = ListPool()
        arena.self_play(tao, pool)
    finally:
        M.MCTS.search.__defaults__ = old
    T = len(pool.items)
    states = np.array([e[0].reshape(-1).numpy() for e in pool.items]).astype(np.int8)
    pis = np.array([e[1].reshape(-1).numpy() for e in pool.items], np.float32)
    zs = np.array([float(e[2][0]) for e in pool.items], np.float32)
    assert pool.items[0][0].shape[0] == 1 and pool.items[0][2].shape == (1,)
    return dict(num_sim=np.int32(num_sim), tao=np.float32(tao), plies=np.int32(T),
                noise=np.array(REC.noise, np.float32), expo=np.array(REC.expo, np.float32),
                actions=np.array(REC.mult_a, np.int32), put_states=states, put_pis=pis, put_z=zs,
                state_shape=np.array(pool.items[0][0].shape, np.int32),
                pi_shape=np.array(pool.items[0][1].shape, np.int32),
                eval_in=np.array(REC.eval_in, np.int8), eval_logits=np.array(REC.eval_logits, np.float32),
                eval_value=np.array(REC.eval_value, np.float32))


def dedup_evals(c):
    """Keep one table row per distinct NN input (the same position is re-evaluated across moves)."""
    seen, keep = set(), []
    for i, row in enumerate(c["eval_in"]):
        k = row.tobytes()
        if k not in seen:
            seen.add(k)
            keep.append(i)
    for k in ("eval_in", "eval_logits", "eval_value"):
        c[k] = c[k][keep]
    return c


def make_selfplay():
    cases = []
    for i, (tao, seed) in enumerate([(1.0, 11), (1.0, 12), (0.5, 13)]):
        cases.append((f"tictactoe_{i}", dedup_evals(selfplay_case(TicTacToeGame, TicTacToeModel, 50, tao, seed, 100))))
    for i, (tao, seed) in enumerate([(1.0, 21), (0.7, 22)]):
        cases.append((f"dab_{i}", dedup_evals(selfplay_case(DotsAndBoxesGame, DotsAndBoxesModel, 40, tao, seed, 101))))
    out = {"tags": np.array([t for t, _ in cases])}
    for t, c in cases:
        print("selfplay", t, "plies", int(c["plies"]), "evals", len(c["eval_in"]))
        for k, v in c.items():
            out[f"{t}__{k}"] = v
    np.savez_compressed(os.path.join(OUT, "selfplay.npz"), **out)


# --------------------------------------------------------------------------- augmentation KATs ----
def make_augment():
    """8-fold symmetry expansion applied by the pools on put (utils/experience_pool.py:27-31,
    utils/share_ring_buffer.py put) = game_cls.get_enhanced_data: utils/data_enhancer.py:13-70 for the
    stone games, examples/dots_and_boxes/game.py:158-320 for dots-and-boxes."""
    rng = np.random.default_rng(99)
    out = {}
    GomokuGame.BOARD_SIZE = 15

    def case(tag, game_cls, cin, h, w, A, n):
        st = rng.integers(-1, 2, size=(n, cin, h, w)).astype(np.float32)
        if game_cls is DotsAndBoxesGame:  # only legal cells are ever non-zero; edges are 0/1
            st[:] = 0
            for b in range(8):
                st[:, 3 * b, :, :3] = rng.integers(0, 2, size=(n, 4, 3))
                st[:, 3 * b + 1, :3, :] = rng.integers(0, 2, size=(n, 3, 4))
                st[:, 3 * b + 2, :3, :3] = rng.integers(-1, 2, size=(n, 3, 3))
        st[:, -1] = rng.choice([-1.0, 1.0], size=(n, 1, 1))
        pi = rng.random((n, A)).astype(np.float32)
        z = rng.choice([-1.0, 0.0, 1.0], size=n).astype(np.float32)
        outs_s, outs_p, outs_z = [], [], []
        for i in range(n):
            res = game_cls.get_enhanced_data(torch.from_numpy(st[i:i + 1]), torch.from_numpy(pi[i:i + 1]),
                                             torch.tensor([z[i]]))
            assert len(res) == 8
            for s_, p_, z_ in res:
                outs_s.append(s_.reshape(-1).numpy().copy())
                outs_p.append(p_.reshape(-1).numpy().copy())
                outs_z.append(float(z_.reshape(-1)[0]))
        out[f"{tag}__in_states"] = st.astype(np.int8)
        out[f"{tag}__in_pis"] = pi
        out[f"{tag}__in_z"] = z
        out[f"{tag}__out_states"] = np.array(outs_s).astype(np.int8)
        out[f"{tag}__out_pis"] = np.array(outs_p, np.float32)
        out[f"{tag}__out_z"] = np.array(outs_z, np.float32)
        print("augment", tag, n, "samples")

    case("tictactoe", TicTacToeGame, 9, 3, 3, 9, 6)
    case("dab", DotsAndBoxesGame, 25, 4, 4, 24, 6)
    case("gomoku15", GomokuGame, 9, 15, 15, 225, 3)
    case("gomoku9", GomokuGame, 9, 9, 9, 81, 4)
    out["tags"] = np.array(["tictactoe", "dab", "gomoku15", "gomoku9"])
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)


# --------------------------------------------------------------------------- learner KATs ---------
def make_loss():
    """training/nn_trainer.py:60-94 `Trainer._calculate_loss` and one `Trainer.train()` (:49-58) on a fixed batch with the
    reference's own model classes: inputs (weights, batch) and outputs (loss, parameters after one SGD step) as data."""
    from training.nn_trainer import Trainer

    out = {}
    rng = np.random.default_rng(17)

    def case(tag, model_cls, cin, h, w, A, B, seed):
        torch.manual_seed(seed)
        model = model_cls()
        st = rng.integers(-1, 2, size=(B, cin, h, w)).astype(np.float32)
        pi = rng.random((B, A)).astype(np.float32)
        pi /= pi.sum(1, keepdims=True)
        z = rng.choice([-1.0, 0.0, 1.0], size=(B, 1)).astype(np.float32)
        batch = (torch.from_numpy(st), torch.from_numpy(pi), torch.from_numpy(z))

        class Pool:
            def sample(self, n, timeout=None):
                return batch

        for k, v in model.state_dict().items():
            out[f"{tag}__w__{k}"] = v.detach().numpy().copy()
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        tr = Trainer(model, Pool(), opt, device=torch.device("cpu"))
        loss = tr._calculate_loss(B)
        out[f"{tag}__loss"] = np.float32(loss.item())
        tr.train(B)
        out[f"{tag}__states"], out[f"{tag}__pis"], out[f"{tag}__z"] = st.astype(np.int8), pi, z
        for k, v in model.state_dict().items():
            a = v.detach().numpy().reshape(-1)
            out[f"{tag}__after_sum__{k}"] = np.float64(a.astype(np.float64).sum())
            out[f"{tag}__after_head__{k}"] = a[:16].copy()
        print("loss", tag, float(loss))

    case("tictactoe", TicTacToeModel, 9, 3, 3, 9, 32, 5)
    case("dab", DotsAndBoxesModel, 25, 4, 4, 24, 32, 6)
    out["tags"] = np.array(["tictactoe", "dab"])
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


# --------------------------------------------------------------------------- QuickModel KAT --------
def formula_weights(state_dict):
    """deterministic weights from a closed formula (so the fixture need not carry 3 MB of floats): the test fills
    alphazero_amd.nets.QuickNet with the same formula"""
    out = {}
    for i, (k, v) in enumerate(state_dict.items()):
        n = v.numel()
        idx = torch.arange(n, dtype=torch.float64)
        if k.endswith("running_var"):
            t = 0.5 + 0.5 * torch.sin(0.37 * idx + i) ** 2
        elif k.endswith("num_batches_tracked"):
            out[k] = v.clone()
            continue
        else:
            t = 0.08 * torch.sin(0.11 * idx * (1 + (i % 7)) + 0.5 * i)
        out[k] = t.to(torch.float32).view_as(v)
    return out


def make_quick():
    """nn_models/quick_model.py QuickModel (default config: 3x3 board, 128 channels, one 5-layer residual block) forward
    on fixed inputs with formula weights: parameter names, shapes and outputs."""
    from nn_models.quick_model import QuickModel

    m = QuickModel().eval()
    m.load_state_dict(formula_weights(m.state_dict()))
    rng = np.random.default_rng(3)
    x = rng.integers(-1, 2, size=(6, 9, 3, 3)).astype(np.float32)
    with torch.no_grad():
        p, v = m(torch.from_numpy(x))
    np.savez_compressed(os.path.join(OUT, "quick.npz"), names=np.array(list(m.state_dict().keys())),
                        shapes=np.array([str(tuple(t.shape)) for t in m.state_dict().values()]),
                        x=x.astype(np.int8), logits=p.numpy(), value=v.numpy())
    print("quick", p.shape, v.shape)


# --------------------------------------------------------------------------- full-board draw -------
def five_free_colouring(n, seed=0):
    """a colouring of the n x n board with ceil(n*n/2) first-player stones and no five in a row of either colour
    (examples/gomoku/game.py:62-89 scans every 5-window in 4 directions), found by local search"""
    rng = np.random.default_rng(seed)
    N = n * n
    windows = []
    for r in range(n):
        for c in range(n):
            for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
                cells = [(r + k * dr, c + k * dc) for k in range(5)]
                if all(0 <= rr < n and 0 <= cc < n for rr, cc in cells):
                    windows.append([rr * n + cc for rr, cc in cells])
    windows = np.array(windows)

    def bad(col):
        sums = col[windows].sum(1)
        return int(((sums == 5) | (sums == 0)).sum())

    col = np.zeros(N, np.int64)
    col[rng.permutation(N)[:(N + 1) // 2]] = 1
    cur = bad(col)
    for _ in range(200000):
        if cur == 0:
            break
        a, b = rng.choice(np.nonzero(col == 1)[0]), rng.choice(np.nonzero(col == 0)[0])
        col[a], col[b] = 0, 1
        nb = bad(col)
        if nb <= cur:
            cur = nb
        else:
            col[a], col[b] = 1, 0
    assert cur == 0, "no five-free colouring found"
    return col


def make_draw():
    """gomoku 9x9 played until the board is full with no five for either side: is_terminal True, _get_winner None,
    terminal_evaluation 0 (examples/gomoku/game.py:47-52, games/base.py:104-114) -- the case random playouts never reach"""
    GomokuGame.BOARD_SIZE = 9
    col = five_free_colouring(9, seed=1)
    p1, p2 = list(np.nonzero(col == 1)[0]), list(np.nonzero(col == 0)[0])
    rng = np.random.default_rng(2)
    rng.shuffle(p1), rng.shuffle(p2)
    script = [int(x) for i in range(41) for x in ((p1[i], p2[i]) if i < 40 else (p1[i],))]
    out = {f"g0_{k}": v for k, v in playout(GomokuGame, rng, script=script).items()}
    assert len(out["g0_actions"]) == 81 and out["g0_terminal"][-1] and not out["g0_terminal"][:-1].any()
    assert not out["g0_winner"].any(), "a five slipped in"
    s = torch.from_numpy(out["g0_states"][-1].astype(np.float32)).view(9, 9)
    assert GomokuGame.is_terminal(s) and GomokuGame._get_winner(s) is None
    out["final_eval_p1"] = np.float32(GomokuGame.terminal_evaluation(s, 1))
    out["final_eval_p2"] = np.float32(GomokuGame.terminal_evaluation(s, -1))
    out["n_games"] = np.int32(1)
    np.savez_compressed(os.path.join(OUT, "rules_gomoku9_draw.npz"), **out)
    GomokuGame.BOARD_SIZE = 15
    print("draw: 81 plies, evaluation", out["final_eval_p1"], out["final_eval_p2"])


# --------------------------------------------------------------------------- shared ring KAT -------
def make_ring():
    """The reference's OWN SharedRingBufferExperiencePool (utils/share_ring_buffer.py:22-127) fed fixed samples through
    put_tupule_experience, in two instalments, with a capacity smaller than the data (wrap-around, overwrite of the oldest
    rows, read index pushed on): the samples and the ring's tensors / indices after each instalment, as data."""
    from utils.share_ring_buffer import SharedRingBufferExperiencePool

    rng = np.random.default_rng(123)
    out = {}
    tags = []

    def case(tag, game_cls, cin, h, w, A, n1, n2, cap):
        n = n1 + n2
        st = rng.integers(-1, 2, size=(n, cin, h, w)).astype(np.float32)
        if game_cls is DotsAndBoxesGame:
            st[:] = 0
            for b in range(8):
                st[:, 3 * b, :, :3] = rng.integers(0, 2, size=(n, 4, 3))
                st[:, 3 * b + 1, :3, :] = rng.integers(0, 2, size=(n, 3, 4))
                st[:, 3 * b + 2, :3, :3] = rng.integers(-1, 2, size=(n, 3, 3))
        st[:, -1] = rng.choice([-1.0, 1.0], size=(n, 1, 1))
        pi = rng.random((n, A)).astype(np.float32)
        z = rng.choice([-1.0, 0.0, 1.0], size=n).astype(np.float32)
        pool = SharedRingBufferExperiencePool((1, cin, h, w), (1, A), game_cls, cap)
        out[f"{tag}__states"], out[f"{tag}__pis"], out[f"{tag}__z"] = st.astype(np.int8), pi, z
        out[f"{tag}__split"], out[f"{tag}__capacity"] = np.int32(n1), np.int32(cap)
        for part, (lo, hi) in enumerate(((0, n1), (n1, n))):
            for i in range(lo, hi):  # exactly what training/self_play.py:122-132 hands the pool
                pool.put_tupule_experience((torch.from_numpy(st[i:i + 1]), torch.from_numpy(pi[i:i + 1]), torch.tensor([z[i]])))
            out[f"{tag}__ring_states_{part}"] = pool._states.numpy().astype(np.int8).copy()
            out[f"{tag}__ring_prior_{part}"] = pool._prior.numpy().copy()
            out[f"{tag}__ring_values_{part}"] = pool._values.numpy().copy()
            out[f"{tag}__idx_{part}"] = np.array([pool._write_idx.value, pool._read_idx.value, pool._size.value,
                                                  pool._total_written.value], np.int64)
        tags.append(tag)
        print("ring", tag, n, "samples into", cap, "rows:", out[f"{tag}__idx_1"])

    GomokuGame.BOARD_SIZE = 9
    case("gomoku9", GomokuGame, 9, 9, 9, 81, 5, 6, 61)   # 88 rows into 61
    case("tictactoe", TicTacToeGame, 9, 3, 3, 9, 7, 9, 100)  # 128 rows into 100, first instalment does not fill it
    case("dab", DotsAndBoxesGame, 25, 4, 4, 24, 4, 5, 40)  # 72 rows into 40: more than one lap
    GomokuGame.BOARD_SIZE = 15
    out["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(OUT, "ring.npz"), **out)


# ------------------------------------------------------------ gomoku under other BOARD_SIZE / WIN_LENGTH constants ----
MNK = [(7, 4), (11, 5), (6, 3), (13, 6), (5, 5), (3, 3), (12, 4), (8, 6)]  # (BOARD_SIZE, WIN_LENGTH): examples/gomoku/game.py:21-22


def gomoku_scripts_nk(n, k):
    def rc(r, c):
        return r * n + c

    S = []
    if n >= k + 1:  # (a second row / column for the other player)
        S.append([x for i in range(k) for x in (rc(0, i), rc(n - 1, i))][:2 * k - 1])                    # horizontal, P1
        S.append([x for i in range(k) for x in (rc(i, n - 1), rc(i, 0))][:2 * k - 1])                    # vertical on the last column
        S.append([x for i in range(k) for x in (rc(n - k + i, n - k + i), rc(0, i))][:2 * k - 1])        # diagonal into the corner
        S.append([x for i in range(k) for x in (rc(i, k - 1 - i), rc(n - 1, i))][:2 * k - 1])            # anti-diagonal from the left edge
    if n >= 2 * k + 1:
        # k + 1 in a row, the gap filled last; P2 scattered on two rows with gaps
        cols = [c for c in range(k + 1) if c != k // 2] + [k // 2]
        p1 = [rc(2, c) for c in cols]
        p2 = [rc(n - 1 - (i % 2), 2 * i) for i in range(k)]
        S.append([x for i in range(k + 1) for x in ((p1[i], p2[i]) if i < k else (p1[i],))])
        # the second player wins with a vertical line; P1 scattered
        S.append([x for i in range(k) for x in (rc(n - 1 - (i % 2), 2 * i), rc(i, 3))])
    return S


class GomokuNModel(BaseModel):
    """a small net of the reference's model contract for an n x n board (the reference's GomokuModel hard-codes 15)"""

    def __init__(self, n):
        super().__init__()
        self._s = torch.nn.Sequential(torch.nn.Conv2d(9, 8, 3, padding=1), torch.nn.ReLU())
        self._p = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(8 * n * n, n * n))
        self._v = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(8 * n * n, 1), torch.nn.Tanh())

    @property
    def shared_layers(self):
        return self._s

    def policy_head(self, x):
        return self._p(x)

    def value_head(self, x):
        return self._v(x)


def make_mnk():
    """the reference's own GomokuGame / MCTS with its two constants changed: rule games (scripted + random) for every pair of
    MNK, and searches on three of them"""
    try:
        for n, k in MNK:
            GomokuGame.BOARD_SIZE, GomokuGame.WIN_LENGTH = n, k
            rules_fixture(f"gomoku{n}k{k}", GomokuGame, gomoku_scripts_nk(n, k), 14, 900 + 16 * n + k)
        rng = np.random.default_rng(11)
        cases = []
        for j, (n, k, specs) in enumerate([(7, 4, [(0, 80, True), (8, 120, True), (20, 60, False)]),
                                           (11, 5, [(0, 60, True), (30, 90, True)]),
                                           (6, 3, [(2, 100, True), (5, 150, False)])]):
            GomokuGame.BOARD_SIZE, GomokuGame.WIN_LENGTH = n, k
            torch.manual_seed(700 + j)
            model = GomokuNModel(n).eval()
            for i, (nm, ns, ud) in enumerate(specs):
                c = search_case(GomokuGame, model, nm, ns, ud, rng, 800 + 10 * j + i)
                cases.append((f"gomoku{n}k{k}_{i}", c))
                print("search", cases[-1][0], "moves", nm, "sims", ns, "maxdepth", c["trace"][:, 0].max())
        out = {"tags": np.array([t for t, _ in cases])}
        for t, c in cases:
            for kk, v in c.items():
                out[f"{t}__{kk}"] = v
        np.savez_compressed(os.path.join(OUT, "search_mnk.npz"), **out)
    finally:
        GomokuGame.BOARD_SIZE, GomokuGame.WIN_LENGTH = 15, 5


DAB_SHAPES = [(2, 2), (4, 4), (5, 5), (2, 3), (3, 2), (4, 5), (5, 4), (3, 5)]  # (BOX_ROWS, BOX_COLS): examples/dots_and_boxes/game.py:24-25


def set_box_shape(r, c):
    """the reference's class constants (examples/dots_and_boxes/game.py:24-34), all of them: the derived ones are computed once,
    in the class body"""
    G = DotsAndBoxesGame
    G.BOX_ROWS, G.BOX_COLS, G.DOT_ROWS, G.DOT_COLS = r, c, r + 1, c + 1
    G.H_EDGE_COUNT, G.V_EDGE_COUNT = (r + 1) * c, r * (c + 1)
    G.NUM_ACTION = G.H_EDGE_COUNT + G.V_EDGE_COUNT
    G.STATE_SHAPE = (3, r + 1, c + 1)


def make_dab_shapes():
    """the reference's own DotsAndBoxesGame / DotsAndBoxesModel / MCTS / ChessArena with BOX_ROWS x BOX_COLS changed: rule
    games for every pair of DAB_SHAPES, searches and one self-play game on some, the 8 symmetries on the square ones"""
    try:
        for r, c in DAB_SHAPES:
            set_box_shape(r, c)
            nh, A = (r + 1) * c, (r + 1) * c + r * (c + 1)
            double = [0, 1, c, c + 1, nh, nh + 2, nh + 1]  # boxes (0,0) and (0,1): every other edge first, then V(0,1) takes both
            rules_fixture(f"dab{r}x{c}", DotsAndBoxesGame, [double, list(range(A)), list(range(A - 1, -1, -1))], 20, 1200 + 8 * r + c)
        rng = np.random.default_rng(12)
        cases = []
        for j, (r, c, specs) in enumerate([(2, 2, [(0, 60, True), (5, 100, True), (9, 60, False)]),
                                           (4, 4, [(0, 80, True), (20, 80, True), (34, 120, False)]),
                                           (2, 3, [(3, 80, True), (12, 80, False)]),
                                           (5, 5, [(10, 60, True), (52, 100, True)]),
                                           (4, 5, [(30, 80, True)])]):
            set_box_shape(r, c)
            torch.manual_seed(720 + j)
            model = DotsAndBoxesModel().eval()
            for i, (nm, ns, ud) in enumerate(specs):
                cs = search_case(DotsAndBoxesGame, model, nm, ns, ud, rng, 900 + 10 * j + i)
                cases.append((f"dab{r}x{c}_{i}", cs))
                print("search", cases[-1][0], "moves", nm, "sims", ns, "maxdepth", cs["trace"][:, 0].max())
        out = {"tags": np.array([t for t, _ in cases])}
        for t, cs in cases:
            for kk, v in cs.items():
                out[f"{t}__{kk}"] = v
        np.savez_compressed(os.path.join(OUT, "search_dab.npz"), **out)

        cases = []
        for r, c, sims, tao, seed in ((2, 2, 30, 1.0, 31), (3, 2, 25, 0.7, 32)):
            set_box_shape(r, c)
            cases.append((f"dab{r}x{c}_0", dedup_evals(selfplay_case(DotsAndBoxesGame, DotsAndBoxesModel, sims, tao, seed, 730 + r))))
        out = {"tags": np.array([t for t, _ in cases])}
        for t, cs in cases:
            print("selfplay", t, "plies", int(cs["plies"]), "evals", len(cs["eval_in"]))
            for kk, v in cs.items():
                out[f"{t}__{kk}"] = v
        np.savez_compressed(os.path.join(OUT, "selfplay_dab.npz"), **out)

        rng = np.random.default_rng(98)
        out, tags = {}, []
        for b in (2, 4, 5):
            set_box_shape(b, b)
            d, A, n = b + 1, 2 * b * (b + 1), 5
            st = np.zeros((n, 25, d, d), np.float32)
            for blk in range(8):  # only legal cells are ever non-zero; edges are 0/1
                st[:, 3 * blk, :, :b] = rng.integers(0, 2, size=(n, d, b))
                st[:, 3 * blk + 1, :b, :] = rng.integers(0, 2, size=(n, b, d))
                st[:, 3 * blk + 2, :b, :b] = rng.integers(-1, 2, size=(n, b, b))
            st[:, -1] = rng.choice([-1.0, 1.0], size=(n, 1, 1))
            pi = rng.random((n, A)).astype(np.float32)
            z = rng.choice([-1.0, 0.0, 1.0], size=n).astype(np.float32)
            outs_s, outs_p, outs_z = [], [], []
            for i in range(n):
                res = DotsAndBoxesGame.get_enhanced_data(torch.from_numpy(st[i:i + 1]), torch.from_numpy(pi[i:i + 1]), torch.tensor([z[i]]))
                assert len(res) == 8
                for s_, p_, z_ in res:
                    outs_s.append(s_.reshape(-1).numpy().copy())
                    outs_p.append(p_.reshape(-1).numpy().copy())
                    outs_z.append(float(z_.reshape(-1)[0]))
            tag = f"dab{b}x{b}"
            tags.append(tag)
            out[f"{tag}__in_states"], out[f"{tag}__in_pis"], out[f"{tag}__in_z"] = st.astype(np.int8), pi, z
            out[f"{tag}__out_states"] = np.array(outs_s).astype(np.int8)
            out[f"{tag}__out_pis"] = np.array(outs_p, np.float32)
            out[f"{tag}__out_z"] = np.array(outs_z, np.float32)
            print("augment", tag, n, "samples")
        out["tags"] = np.array(tags)
        np.savez_compressed(os.path.join(OUT, "augment_dab.npz"), **out)
        # a non-square board has no symmetries in the reference: record what get_enhanced_data does there
        set_box_shape(2, 3)
        try:
            DotsAndBoxesGame.get_enhanced_data(torch.zeros(1, 25, 3, 4), torch.zeros(1, 17), torch.zeros(1))
            print("augment dab2x3: no error")
        except Exception as ex:  # noqa: BLE001
            print("augment dab2x3 raises", type(ex).__name__, ex)
    finally:
        set_box_shape(3, 3)


def set_history_len(h):
    """the reference's HISTORY_LEN (config/basic.py:10 _HISTORY_LEN -> :39-40): every module that did ``from config import *``
    holds its own binding, all read at call time (utils/history_manager.py:21,38,52; the models' __init__)"""
    for mod in list(sys.modules.values()):
        if mod is not None and "HISTORY_LEN" in getattr(mod, "__dict__", {}) and getattr(mod, "__name__", "") != __name__:
            mod.HISTORY_LEN = h


class GomokuHModel(BaseModel):
    """GomokuNModel with the input planes of a given HISTORY_LEN"""

    def __init__(self, n, h):
        super().__init__()
        self._s = torch.nn.Sequential(torch.nn.Conv2d(h + 1, 8, 3, padding=1), torch.nn.ReLU())
        self._p = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(8 * n * n, n * n))
        self._v = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(8 * n * n, 1), torch.nn.Tanh())

    @property
    def shared_layers(self):
        return self._s

    def policy_head(self, x):
        return self._p(x)

    def value_head(self, x):
        return self._v(x)


def make_history():
    """the reference's own MCTS / ChessArena / models with HISTORY_LEN changed: searches and self-play games whose network inputs
    and samples have HISTORY_LEN * C + 1 planes.  Tags carry the value: <game>_h<HISTORY_LEN>_<i>."""
    try:
        rng = np.random.default_rng(13)
        cases = []
        for j, (h, game_cls, mk, tag, specs) in enumerate([
                (3, TicTacToeGame, lambda: TicTacToeModel(), "tictactoe", [(0, 50, True), (4, 80, True), (6, 50, False)]),
                (1, DotsAndBoxesGame, lambda: DotsAndBoxesModel(), "dab", [(0, 80, True), (15, 80, True), (20, 120, False)]),
                (5, GomokuGame, lambda: GomokuHModel(9, 5), "gomoku9", [(0, 80, True), (12, 100, True), (30, 60, False)]),
                (2, GomokuGame, lambda: GomokuModel(), "gomoku15", [(6, 40, True)])]):
            set_history_len(h)
            GomokuGame.BOARD_SIZE = 9 if tag == "gomoku9" else 15
            torch.manual_seed(740 + j)
            model = mk().eval()
            for i, (nm, ns, ud) in enumerate(specs):
                c = search_case(game_cls, model, nm, ns, ud, rng, 950 + 10 * j + i)
                assert c["history"].size == h * c["root_state"].size
                cases.append((f"{tag}_h{h}_{i}", c))
                print("search", cases[-1][0], "moves", nm, "sims", ns, "input", c["eval_in"].shape)
        out = {"tags": np.array([t for t, _ in cases])}
        for t, c in cases:
            for kk, v in c.items():
                out[f"{t}__{kk}"] = v
        np.savez_compressed(os.path.join(OUT, "search_hist.npz"), **out)

        cases = []
        for h, game_cls, model_cls, tag, sims, tao, seed in ((3, TicTacToeGame, TicTacToeModel, "tictactoe", 40, 1.0, 41),
                                                           (2, DotsAndBoxesGame, DotsAndBoxesModel, "dab", 30, 0.8, 42),
                                                           (1, TicTacToeGame, TicTacToeModel, "tictactoe", 40, 1.0, 43)):
            set_history_len(h)
            cases.append((f"{tag}_h{h}_0", dedup_evals(selfplay_case(game_cls, model_cls, sims, tao, seed, 750 + h))))
        out = {"tags": np.array([t for t, _ in cases])}
        for t, c in cases:
            print("selfplay", t, "plies", int(c["plies"]), "evals", len(c["eval_in"]), "state", c["state_shape"])
            for kk, v in c.items():
                out[f"{t}__{kk}"] = v
        np.savez_compressed(os.path.join(OUT, "selfplay_hist.npz"), **out)
    finally:
        set_history_len(8)
        GomokuGame.BOARD_SIZE = 15


if __name__ == "__main__":
    which = sys.argv[1:] or ["rules", "search", "selfplay", "augment", "loss", "quick"]
    if "quick" in which:
        make_quick()
    if "loss" in which:
        make_loss()
    if "rules" in which:
        make_rules()
    if "search" in which:
        make_search()
    if "selfplay" in which:
        make_selfplay()
    if "augment" in which:
        make_augment()
    if "draw" in which:  # (added in round 3; not in the default list, so the older fixtures are not rewritten)
        make_draw()
    if "ring" in which:
        make_ring()
    if "mnk" in which:  # gomoku under other BOARD_SIZE / WIN_LENGTH constants
        make_mnk()
    if "dabshapes" in which:  # dots-and-boxes under other BOX_ROWS / BOX_COLS constants
        make_dab_shapes()
    if "history" in which:  # other values of HISTORY_LEN
        make_history()
