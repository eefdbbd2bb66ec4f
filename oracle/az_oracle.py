"""ctypes front end of oracle/libaz_oracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package ``alphazero_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libaz_oracle.so")

GAME_TICTACTOE, GAME_DAB, GAME_GOMOKU = 0, 1, 2
GAME_IDS = {"tictactoe": GAME_TICTACTOE, "dab": GAME_DAB, "gomoku": GAME_GOMOKU}

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int,
                      C.POINTER(C.c_float))

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "az_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libaz_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.az_oracle_search.restype = C.c_int
        _lib.az_oracle_self_play.restype = C.c_int
        _lib.az_oracle_status.restype = C.c_int
        _lib.az_oracle_next_state.restype = C.c_int
        _lib.az_oracle_powf.restype = C.c_float
        _lib.az_oracle_powf.argtypes = [C.c_float, C.c_float]
    return _lib


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


class win_length:
    """``with win_length(k):`` the oracle's rule code runs with the reference's WIN_LENGTH constant (examples/gomoku/game.py:22)
    set to k (3 .. 6) -- process-wide, restored on exit.  Test infrastructure, like everything here."""

    def __init__(self, k: int):
        self.k, self.old = int(k), None

    def __enter__(self):
        if not 3 <= self.k <= 6:
            raise ValueError("WIN_LENGTH 3 .. 6")
        self.old = lib().az_oracle_set_win_length(self.k)
        return self

    def __exit__(self, *exc):
        lib().az_oracle_set_win_length(self.old)
        return False


class box_shape:
    """``with box_shape(rows, cols):`` the oracle's dots-and-boxes code runs with the reference's BOX_ROWS / BOX_COLS constants
    (examples/dots_and_boxes/game.py:24-25) set to rows x cols (2 .. 5 each) -- process-wide, restored on exit."""

    def __init__(self, rows: int, cols: int):
        self.r, self.c, self.old = int(rows), int(cols), None

    def __enter__(self):
        if not (2 <= self.r <= 5 and 2 <= self.c <= 5):
            raise ValueError("BOX_ROWS, BOX_COLS 2 .. 5")
        self.old = lib().az_oracle_set_box_shape(self.r, self.c)
        return self

    def __exit__(self, *exc):
        lib().az_oracle_set_box_shape(self.old // 8, self.old % 8)
        return False


class history_len:
    """``with history_len(h):`` the oracle runs with the reference's HISTORY_LEN constant (config/basic.py:10) set to h (1 .. 8) --
    process-wide, restored on exit."""

    def __init__(self, h: int):
        self.h, self.old = int(h), None

    def __enter__(self):
        if not 1 <= self.h <= 8:
            raise ValueError("HISTORY_LEN 1 .. 8")
        self.old = lib().az_oracle_set_history_len(self.h)
        return self

    def __exit__(self, *exc):
        lib().az_oracle_set_history_len(self.old)
        return False


def dims(game: int, board_size: int = 15) -> dict:
    out = (C.c_int * 6)()
    if lib().az_oracle_dims(game, board_size, out) != 0:
        raise ValueError("bad game")
    return dict(A=out[0], C=out[1], H=out[2], W=out[3], numel=out[4], cin=out[5])


def hash_eval(scale: float = 4.0):
    """(fn, ctx) for the built-in synthetic evaluator; ctx keeps the scale alive."""
    s = C.c_float(scale)
    fn = C.cast(lib().az_oracle_eval_hash, EVAL_FN)
    return fn, C.cast(C.pointer(s), C.c_void_p), s


def torch_eval(model, shape):
    """Evaluator that runs a torch module on CPU with batch 1, as the reference does
    (core/MCTS_alphazero.py:160-162)."""
    import torch

    cin, h, w = shape

    def _cb(_ctx, x, n_in, logits, n_action, value):
        arr = np.ctypeslib.as_array(x, shape=(n_in,))
        with torch.inference_mode():
            p, v = model(torch.from_numpy(arr.copy()).view(1, cin, h, w))
        np.ctypeslib.as_array(logits, shape=(n_action,))[:] = p.reshape(-1).float().numpy()
        value[0] = float(v.reshape(-1)[0].item())

    return EVAL_FN(_cb)


def table_eval(table: dict):
    """Evaluator that looks (logits, value) up by the bytes of the NN input (fixtures)."""

    def _cb(_ctx, x, n_in, logits, n_action, value):
        key = np.ctypeslib.as_array(x, shape=(n_in,)).tobytes()
        lg, v = table[key]
        np.ctypeslib.as_array(logits, shape=(n_action,))[:] = lg
        value[0] = v

    return EVAL_FN(_cb)


def search(game, board_size, root_state, root_player, history, num_sim, eval_fn, ctx=None, c_puct=1.0,
           epsilon=0.25, noise=None, prior_override=None, trace_stride=0):
    d = dims(game, board_size)
    A = d["A"]
    root_state = np.ascontiguousarray(root_state, np.float32).reshape(-1)
    history = np.ascontiguousarray(history, np.float32).reshape(-1)
    assert root_state.size == d["numel"] and history.size == (d["cin"] - 1) * d["H"] * d["W"]
    noise = None if noise is None else np.ascontiguousarray(noise, np.float32).reshape(-1)
    po = None if prior_override is None else np.ascontiguousarray(prior_override, np.float32)
    pi = np.zeros(A, np.float32)
    visits = np.zeros(A, np.int32)
    child_q = np.zeros(A, np.float64)
    root_prior = np.zeros(A, np.float32)
    trace = np.full((num_sim, max(trace_stride, 1)), -1, np.int32)
    leaf_prior = np.zeros((num_sim, A), np.float32)
    stats = np.zeros(4, np.float64)
    rc = lib().az_oracle_search(
        C.c_int(game), C.c_int(board_size), _p(root_state, _f32p), C.c_int(root_player), _p(history, _f32p),
        C.c_int(num_sim), C.c_float(c_puct), C.c_float(epsilon), _p(noise, _f32p), _p(po, _f32p), eval_fn, ctx,
        _p(pi, _f32p), _p(visits, _i32p), _p(child_q, _f64p), _p(root_prior, _f32p),
        _p(trace, _i32p) if trace_stride else None, C.c_int(trace_stride), _p(leaf_prior, _f32p),
        _p(stats, _f64p))
    if rc != 0:
        raise RuntimeError(f"az_oracle_search failed: {rc}")
    return dict(pi=pi, visits=visits, child_q=child_q, root_prior=root_prior, trace=trace,
                leaf_prior=leaf_prior, stats=stats)


def search_wave(game, board_size, root_state, root_player, history, num_sim, leaves_per_wave, eval_fn, ctx=None,
                virtual_loss=0.3, c_puct=1.0, epsilon=0.25, noise=None):
    """the search with virtual loss / in-tree leaf batching (az_oracle_search_wave); leaves_per_wave == 1 is search()"""
    d = dims(game, board_size)
    A = d["A"]
    root_state = np.ascontiguousarray(root_state, np.float32).reshape(-1)
    history = np.ascontiguousarray(history, np.float32).reshape(-1)
    noise = None if noise is None else np.ascontiguousarray(noise, np.float32).reshape(-1)
    pi, visits = np.zeros(A, np.float32), np.zeros(A, np.int32)
    child_q, root_prior, stats = np.zeros(A, np.float64), np.zeros(A, np.float32), np.zeros(5, np.float64)
    fn = lib().az_oracle_search_wave
    fn.restype = C.c_int
    rc = fn(C.c_int(game), C.c_int(board_size), _p(root_state, _f32p), C.c_int(root_player), _p(history, _f32p),
            C.c_int(num_sim), C.c_int(leaves_per_wave), C.c_double(virtual_loss), C.c_float(c_puct), C.c_float(epsilon),
            _p(noise, _f32p), eval_fn, ctx, _p(pi, _f32p), _p(visits, _i32p), _p(child_q, _f64p), _p(root_prior, _f32p),
            _p(stats, _f64p))
    if rc != 0:
        raise RuntimeError(f"az_oracle_search_wave failed: {rc}")
    return dict(pi=pi, visits=visits, child_q=child_q, root_prior=root_prior, stats=stats)


def self_play(game, board_size, num_sim, tao, expo, eval_fn, ctx=None, noise=None, c_puct=1.0,
              epsilon=0.25, max_plies=None, want_states=True):
    d = dims(game, board_size)
    A = d["A"]
    nin = d["cin"] * d["H"] * d["W"]
    expo = np.ascontiguousarray(expo, np.float32)
    max_plies = max_plies or expo.shape[0]
    assert expo.shape[0] >= max_plies and expo.shape[1] == A
    noise = None if noise is None else np.ascontiguousarray(noise, np.float32)
    actions = np.zeros(max_plies, np.int32)
    players = np.zeros(max_plies, np.int32)
    states = np.zeros((max_plies, nin), np.float32) if want_states else None
    pis = np.zeros((max_plies, A), np.float32)
    z = np.zeros(max_plies, np.float32)
    result = C.c_float(0)
    stats = np.zeros(4, np.float64)
    rc = lib().az_oracle_self_play(
        C.c_int(game), C.c_int(board_size), C.c_int(num_sim), C.c_float(c_puct), C.c_float(epsilon),
        C.c_double(tao), _p(noise, _f32p), _p(expo, _f32p), C.c_int(max_plies), eval_fn, ctx,
        _p(actions, _i32p), _p(players, _i32p), _p(states, _f32p), _p(pis, _f32p), _p(z, _f32p),
        C.byref(result), _p(stats, _f64p))
    if rc <= 0:
        raise RuntimeError(f"az_oracle_self_play failed: {rc}")
    T = rc
    return dict(plies=T, actions=actions[:T], players=players[:T],
                nn_states=None if states is None else states[:T], pis=pis[:T], z=z[:T],
                result=float(result.value), stats=stats)


def next_state(game, board_size, state, action, player):
    d = dims(game, board_size)
    state = np.ascontiguousarray(state, np.float32).reshape(-1)
    out = np.zeros(d["numel"], np.float32)
    np_ = C.c_int(0)
    rc = lib().az_oracle_next_state(C.c_int(game), C.c_int(board_size), _p(state, _f32p), C.c_int(action),
                                    C.c_int(player), _p(out, _f32p), C.byref(np_))
    if rc != 0:
        raise ValueError("illegal move" if rc == -2 else "action out of range")
    return out, int(np_.value)


def legal_mask(game, board_size, state):
    d = dims(game, board_size)
    state = np.ascontiguousarray(state, np.float32).reshape(-1)
    m = np.zeros(d["A"], np.float32)
    lib().az_oracle_legal_mask(C.c_int(game), C.c_int(board_size), _p(state, _f32p), _p(m, _f32p))
    return m


def status(game, board_size, state):
    """-> (is_terminal, winner) with winner in {+1, -1, 0 (None)}."""
    state = np.ascontiguousarray(state, np.float32).reshape(-1)
    w = C.c_int(0)
    t = lib().az_oracle_status(C.c_int(game), C.c_int(board_size), _p(state, _f32p), C.byref(w))
    return bool(t), int(w.value)


def get_prior(game, board_size, logits, mask, noise=None, epsilon=0.25):
    logits = np.ascontiguousarray(logits, np.float32).reshape(-1)
    mask = np.ascontiguousarray(mask, np.float32).reshape(-1)
    noise = None if noise is None else np.ascontiguousarray(noise, np.float32).reshape(-1)
    out = np.zeros_like(logits)
    lib().az_oracle_get_prior(C.c_int(game), C.c_int(board_size), _p(logits, _f32p), _p(mask, _f32p),
                              _p(noise, _f32p), C.c_float(epsilon), _p(out, _f32p))
    return out


def powf(x, y):
    return float(lib().az_oracle_powf(C.c_float(x), C.c_float(y)))


def philox4x32_10(counter, key):
    """the Philox4x32-10 block function (az_oracle_philox4x32_10)"""
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in counter])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    out = (C.c_uint32 * 4)()
    lib().az_oracle_philox4x32_10(c, k, out)
    return [int(x) for x in out]


def draw_rng(seed, game_id, ply, n_action, alpha=0.3, want_noise=True, want_expo=True):
    """the per-game random rows of game ``game_id`` at ply ``ply`` (az_oracle_draw_rng == the engine's az_draw_rng):
    -> (noise (A,) Dirichlet(alpha) or None, expo (A,) Exponential(1) or None)"""
    noise = np.zeros(n_action, np.float32) if want_noise else None
    expo = np.zeros(n_action, np.float32) if want_expo else None
    fn = lib().az_oracle_draw_rng
    fn.restype = C.c_int
    rc = fn(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_int64(int(game_id)), C.c_int(ply), C.c_int(n_action), C.c_float(alpha),
            _p(noise, _f32p), _p(expo, _f32p))
    if rc != 0:
        raise ValueError(f"az_oracle_draw_rng failed: {rc}")
    return noise, expo
