"""Pins the CPU oracle (oracle/az_oracle.c) to the reference: every golden vector in tests/golden/
was produced by running the reference itself (tests/golden/make_golden.py).

Bars: rules, visit counts, per-simulation descent paths, chosen moves, z and NN-input tensors are
bit-exact; priors / pi / Q are compared at 1e-5 relative (north_star), because softmax's expf and the
order of Tensor.sum() live inside torch, not in the reference's sources.
"""
import os

import numpy as np
import pytest

from .conftest import DAB_RULES, GOLDEN, MNK_RULES, game_of, load_cases, reference_constants

# gomoku9_draw: a scripted 81-ply game that fills the board with no five for either side (the reference says terminal, no
# winner, evaluation 0: examples/gomoku/game.py:47-52) -- the case random playouts never reach
RULES = [("tictactoe", 0, 3), ("dab", 1, 3), ("gomoku15", 2, 15), ("gomoku9", 2, 9), ("gomoku9_draw", 2, 9)]


@pytest.mark.parametrize("name,game,n", RULES + MNK_RULES + DAB_RULES)
def test_rules_kat(oracle, name, game, n):
    """(the gomoku<N>k<K> fixtures: the reference's GomokuGame with its BOARD_SIZE / WIN_LENGTH constants set to N / K; dab<R>x<C>:
    its DotsAndBoxesGame with BOX_ROWS x BOX_COLS = R x C)"""
    with reference_constants(oracle, name):
        _rules_kat(oracle, name, game, n)


def _rules_kat(oracle, name, game, n):
    z = np.load(os.path.join(GOLDEN, f"rules_{name}.npz"))
    d = oracle.dims(game, n)
    for g in range(int(z["n_games"])):
        s = np.zeros(d["numel"], np.float32)
        assert np.array_equal(oracle.legal_mask(game, n, s), z[f"g{g}_mask0"])
        player = 1
        for k, a in enumerate(z[f"g{g}_actions"]):
            s, player = oracle.next_state(game, n, s, int(a), player)
            assert np.array_equal(s.astype(np.int8), z[f"g{g}_states"][k]), (g, k)
            assert player == z[f"g{g}_next_players"][k]
            assert np.array_equal(oracle.legal_mask(game, n, s), z[f"g{g}_masks"][k])
            term, winner = oracle.status(game, n, s)
            assert term == bool(z[f"g{g}_terminal"][k]) and winner == z[f"g{g}_winner"][k], (g, k)
    if name.endswith("_draw"):
        assert term and winner == 0 and k == d["A"] - 1 and not z["g0_terminal"][:-1].any()
        assert float(z["final_eval_p1"]) == 0.0 == float(z["final_eval_p2"])
        return
    # error behaviour: occupied / out-of-range raise ValueError in the reference
    s = np.zeros(d["numel"], np.float32)
    s, _ = oracle.next_state(game, n, s, 0, 1)
    for a, key in ((0, "err_occupied"), (-1, "err_neg"), (10 ** 6, "err_big")):
        raised = False
        try:
            oracle.next_state(game, n, s, a, -1)
        except ValueError:
            raised = True
        assert raised == bool(z[key]), key


def _table(c):
    return {c["eval_in"][i].astype(np.float32).tobytes(): (c["eval_logits"][i], float(c["eval_value"][i]))
            for i in range(len(c["eval_in"]))}


SEARCH = {**load_cases("search.npz"), **load_cases("search_mnk.npz"), **load_cases("search_dab.npz"), **load_cases("search_hist.npz")}


@pytest.mark.parametrize("tag", sorted(SEARCH))
def test_search_kat(oracle, tag):
    c = SEARCH[tag]
    game, n = game_of(tag)
    ev = oracle.table_eval(_table(c))  # KeyError here <=> the oracle built a different NN input
    noise = c["noise"] if c["noise"].size else None
    D = c["trace"].shape[1]
    with reference_constants(oracle, tag):
        r = oracle.search(game, n, c["root_state"].astype(np.float32), int(c["root_player"]),
                          c["history"].astype(np.float32), int(c["num_sim"]), ev, noise=noise, trace_stride=D)
    assert np.array_equal(r["visits"], c["visits"])           # bit-exact decisions
    assert np.array_equal(r["trace"], c["trace"])             # every simulation took the same path
    assert np.array_equal(r["pi"], c["pi"])                   # visits/sum: exact
    np.testing.assert_allclose(r["root_prior"], c["root_prior"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(r["child_q"], c["child_q"], rtol=1e-5, atol=1e-9)
    lp = r["leaf_prior"][r["leaf_prior"].sum(1) > 0]
    np.testing.assert_allclose(lp, c["priors"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", sorted(SEARCH))
def test_search_kat_reference_priors(oracle, tag):
    """Same, with the reference's own final priors fed back in: then Q is bit-exact too, which pins
    the tree arithmetic (PUCT dtype chain, f64 running mean, sign-aware backup) on its own."""
    c = SEARCH[tag]
    game, n = game_of(tag)
    ev = oracle.table_eval(_table(c))
    noise = c["noise"] if c["noise"].size else None
    with reference_constants(oracle, tag):
        r = oracle.search(game, n, c["root_state"].astype(np.float32), int(c["root_player"]),
                          c["history"].astype(np.float32), int(c["num_sim"]), ev, noise=noise,
                          prior_override=c["priors"], trace_stride=c["trace"].shape[1])
    assert np.array_equal(r["visits"], c["visits"])
    assert np.array_equal(r["trace"], c["trace"])
    assert np.array_equal(r["child_q"], c["child_q"])
    assert np.array_equal(r["root_prior"], c["root_prior"])


SELFPLAY = {**load_cases("selfplay.npz"), **load_cases("selfplay_dab.npz"), **load_cases("selfplay_hist.npz")}


@pytest.mark.parametrize("tag", sorted(SELFPLAY))
def test_selfplay_kat(oracle, tag):
    c = SELFPLAY[tag]
    game, n = game_of(tag)
    ev = oracle.table_eval(_table(c))
    with reference_constants(oracle, tag):
        r = oracle.self_play(game, n, int(c["num_sim"]), float(c["tao"]), c["expo"], ev, noise=c["noise"])
    T = int(c["plies"])
    assert r["plies"] == T
    assert np.array_equal(r["actions"], c["actions"])
    # the reference puts the last position first (training/self_play.py:120-139)
    assert np.array_equal(r["nn_states"][::-1].astype(np.int8), c["put_states"])
    assert np.array_equal(r["pis"][::-1], c["put_pis"])
    assert np.array_equal(r["z"][::-1], c["put_z"])
    assert tuple(c["state_shape"][:1]) == (1,) and tuple(c["pi_shape"]) == (1, r["pis"].shape[1])


def test_expf_logf_accuracy(oracle):
    import math

    xs = np.linspace(-80, 0, 4001, dtype=np.float32)
    lg = np.zeros(2, np.float32)
    for x in xs[::40]:
        # softmax of [0, x] exercises az_expf(x)
        p = oracle.get_prior(0, 3, np.array([0, x] + [-1e9] * 7, np.float32), np.ones(9, np.float32))
        want = math.exp(float(x)) / (1 + math.exp(float(x)))
        assert abs(p[1] - want) <= 3e-7 * want + 1e-38
    for x, y in [(0.5, 2.0), (0.02, 10.0), (0.9, 1.0 / 0.7), (1.0, 3.0), (0.0, 2.0), (0.3, 1.0)]:
        assert abs(oracle.powf(x, y) - x ** y) <= 2e-5 * (x ** y) + 1e-30


@pytest.mark.parametrize("tag", ["tictactoe", "dab", "gomoku15", "gomoku9", "dab2x2", "dab4x4", "dab5x5"])
def test_augmentation_kat(tag):
    """oracle/augment.py (numpy) == the reference's get_enhanced_data on every golden sample, bit-exact (dab<B>x<B>: with
    BOX_ROWS = BOX_COLS = B; a non-square board has no symmetries -- the reference raises, make_golden.py records that)"""
    from oracle import augment as AUG

    z = np.load(os.path.join(GOLDEN, "augment_dab.npz" if "x" in tag else "augment.npz"))
    game, _ = game_of(tag)
    st, pi, zz = z[f"{tag}__in_states"].astype(np.float32), z[f"{tag}__in_pis"], z[f"{tag}__in_z"]
    for i in range(st.shape[0]):
        outs, outp, outz = AUG.augment(game, st[i], pi[i], zz[i])
        for t in range(8):
            assert np.array_equal(outs[t].reshape(-1).astype(np.int8), z[f"{tag}__out_states"][8 * i + t]), (i, t)
            assert np.array_equal(outp[t], z[f"{tag}__out_pis"][8 * i + t]), (i, t)
            assert outz[t] == z[f"{tag}__out_z"][8 * i + t]


# ---------------------------------------------------------------- per-game random rows (az_draw_rng) ----
PHILOX_KAT = [  # Random123 known-answer vectors for philox4x32-10: (counter, key, output)
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def test_philox_known_answers_oracle_and_library(oracle):
    """the block function behind the per-game rows, in the oracle's restatement and in the product library's host export,
    against the published Random123 vectors (the generator is a third-party algorithm: Salmon et al., SC'11)"""
    import ctypes as C

    from alphazero_amd.engine import load_library

    lib = load_library()
    for ctr, key, want in PHILOX_KAT:
        assert oracle.philox4x32_10(ctr, key) == want
        out = (C.c_uint32 * 4)()
        assert lib.az_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out) == 0
        assert list(out) == want


def test_per_game_rows_are_a_function_of_seed_game_and_ply_with_the_right_distribution(oracle):
    A = 9
    n0, e0 = oracle.draw_rng(7, 3, 2, A)
    n1, e1 = oracle.draw_rng(7, 3, 2, A)
    assert np.array_equal(n0, n1) and np.array_equal(e0, e1)
    for other in ((8, 3, 2), (7, 4, 2), (7, 3, 3), (7, 3 + 2 ** 32, 2)):
        n2, e2 = oracle.draw_rng(*other, A)
        assert not np.array_equal(n0, n2) and not np.array_equal(e0, e2)
    # a longer row starts with other values only through the normalisation: the Exponential row is a prefix
    assert np.array_equal(oracle.draw_rng(7, 3, 2, 24)[1][:A], e0)
    ns = np.stack([oracle.draw_rng(1, g, 0, A)[0] for g in range(20000)])
    es = np.stack([oracle.draw_rng(1, g, 0, A)[1] for g in range(20000)])
    assert np.all(es > 0) and abs(es.mean() - 1.0) < 0.01 and abs(es.var() - 1.0) < 0.03  # Exponential(1)
    assert np.allclose(ns.sum(1), 1.0, atol=1e-5) and np.all(ns >= 0)
    a0 = 0.3 * A  # Dirichlet(0.3 * 1_A): mean 1/A, variance (1/A)(1 - 1/A) / (a0 + 1)
    assert np.all(np.abs(ns.mean(0) - 1.0 / A) < 0.005)
    assert abs(ns.var(0).mean() - (1.0 / A) * (1 - 1.0 / A) / (a0 + 1)) < 0.001
    ref = np.random.default_rng(0).dirichlet([0.3] * A, 20000)[:, 0]
    for q in (0.25, 0.5, 0.9, 0.99):  # the marginal is Beta(0.3, 2.4): quantiles against numpy's sampler
        assert abs(np.quantile(ns[:, 0], q) - np.quantile(ref, q)) < 0.02
