"""ctypes binding of libaz_engine.so (C ABI: include/az_engine.h) on torch device tensors.

PyTorch is plumbing here: it owns the caller-side device buffers and the stream.  Every compute step
is a hand-written HIP kernel inside the library; if the library is missing this module raises -- there
is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional
from dataclasses import dataclass

import torch

from . import _build

GAME_TICTACTOE, GAME_DAB, GAME_GOMOKU = 0, 1, 2
NN_F32, NN_BF16, NN_F16 = 0, 1, 2
_NN_DTYPES = {torch.float32: NN_F32, torch.bfloat16: NN_BF16, torch.float16: NN_F16}


class AzConfig(C.Structure):
    _fields_ = [("game", C.c_int32), ("board_size", C.c_int32), ("n_slots", C.c_int32), ("max_sims", C.c_int32),
                ("c_puct", C.c_float), ("epsilon", C.c_float), ("nn_dtype", C.c_int32), ("device", C.c_int32),
                ("edge_cap", C.c_int32), ("leaves_per_step", C.c_int32), ("win_length", C.c_int32), ("board_cols", C.c_int32),
                ("history_len", C.c_int32)]


class AzDims(C.Structure):
    _fields_ = [("n_action", C.c_int32), ("state_channels", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("nn_channels", C.c_int32), ("max_plies", C.c_int32), ("state_words", C.c_int32),
                ("edge_cap", C.c_int32), ("node_cap", C.c_int32), ("depth_cap", C.c_int32), ("n_slots", C.c_int32),
                ("leaves_per_step", C.c_int32), ("device_bytes", C.c_int64)]


class AzStats(C.Structure):
    _fields_ = [("sims", C.c_uint64), ("sum_depth", C.c_uint64), ("sum_children", C.c_uint64),
                ("evals", C.c_uint64), ("terminals", C.c_uint64), ("expanded_edges", C.c_uint64),
                ("sum_touched", C.c_uint64)]


EXPORTS = ["az_create", "az_destroy", "az_get_dims", "az_last_error", "az_check", "az_reset_games",
           "az_set_position", "az_begin_search", "az_select_gather", "az_expand_backup", "az_expand_select", "az_root_policy",
           "az_root_nn_state", "az_play_moves", "az_game_info", "az_emit_samples", "az_read_stats",
           "az_reset_stats", "az_eval_hash", "az_tower_forward", "az_tower_last_error", "az_heads_forward", "az_heads_feat_stride", "az_augment_samples", "az_leaf_buffers", "az_compact_leaves", "az_copy_to_host",
           "az_tower_heads_forward_static", "az_static_max_boards", "az_tower_forward_f32", "az_heads_forward_f32",
           "az_select_wave", "az_expand_wave", "az_search_remaining", "az_set_virtual_loss", "az_smallnet_forward_f32",
           "az_quick_tower_forward_f32", "az_quick_heads_forward_f32", "az_quick_stash_bytes", "az_quick_tower_forward", "az_quick_heads_forward",
           "az_smallnet_head_floats", "az_draw_rng", "az_philox4x32_10", "az_tower_forward_h2", "az_tower_forward_h1", "az_tower_forward_f32w"]

_lib = None


def load_library() -> C.CDLL:
    """Load the in-tree HIP library; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m alphazero_amd._build` (hipcc, gfx950). "
            "alphazero_amd has no CPU fallback.")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.az_create.argtypes = [C.POINTER(AzConfig), C.POINTER(vp)]
    lib.az_destroy.argtypes = [vp]
    lib.az_get_dims.argtypes = [vp, C.POINTER(AzDims)]
    lib.az_last_error.argtypes = [vp]
    lib.az_last_error.restype = C.c_char_p
    lib.az_check.argtypes = [vp, vp]
    lib.az_reset_games.argtypes = [vp, vp, vp]
    lib.az_set_position.argtypes = [vp, C.c_int32, vp, C.c_int32, vp]
    lib.az_begin_search.argtypes = [vp, vp]
    lib.az_select_gather.argtypes = [vp, vp, vp]
    lib.az_expand_backup.argtypes = [vp, vp, vp, vp, vp]
    lib.az_expand_select.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.az_root_policy.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.az_root_nn_state.argtypes = [vp, vp, vp]
    lib.az_play_moves.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.az_game_info.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.az_emit_samples.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, vp]
    lib.az_read_stats.argtypes = [vp, C.POINTER(AzStats), vp]
    lib.az_reset_stats.argtypes = [vp, vp]
    lib.az_eval_hash.argtypes = [vp, vp, C.c_float, vp, vp, vp]
    lib.az_augment_samples.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int64, vp]
    lib.az_tower_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp]
    lib.az_heads_forward.argtypes = [C.c_int32, C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp]
    lib.az_heads_feat_stride.argtypes = [C.c_int32]
    lib.az_heads_feat_stride.restype = C.c_int32
    lib.az_compact_leaves.argtypes = [vp, vp]
    lib.az_copy_to_host.argtypes = [vp, vp, C.c_uint64, C.c_int32, vp]
    lib.az_leaf_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.az_static_max_boards.argtypes = [C.c_int32]
    lib.az_static_max_boards.restype = C.c_int32
    lib.az_tower_heads_forward_static.argtypes = [C.c_int32] * 4 + [vp] * 14 + [C.c_float, vp, vp, vp]
    lib.az_tower_forward_f32.argtypes = [C.c_int32] * 4 + [vp] * 10 + [C.c_int32, vp]
    lib.az_tower_forward_f32w.argtypes = [C.c_int32] * 4 + [vp] * 10 + [C.c_int32, vp]
    lib.az_heads_forward_f32.argtypes = [C.c_int32, C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp]
    lib.az_smallnet_forward_f32.argtypes = [C.c_int32, C.c_int32] + [vp] * 12
    lib.az_smallnet_head_floats.argtypes = [C.c_int32]
    lib.az_quick_stash_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.az_quick_stash_bytes.restype = C.c_int64
    lib.az_quick_tower_forward_f32.argtypes = [C.c_int32] * 5 + [vp] * 10 + [C.c_int32, vp, C.c_int64, vp]
    lib.az_quick_heads_forward_f32.argtypes = [C.c_int32, C.c_int32, vp, C.c_int32] + [vp] * 7 + [C.c_float, vp, vp, vp]
    lib.az_quick_tower_forward.argtypes = [C.c_int32] * 5 + [vp] * 10 + [C.c_int32, vp]
    lib.az_quick_heads_forward.argtypes = [C.c_int32, C.c_int32, vp, C.c_int32] + [vp] * 7 + [C.c_float, vp, vp, vp]
    lib.az_smallnet_head_floats.restype = C.c_int32
    lib.az_select_wave.argtypes = [vp, vp, C.c_int32, vp]
    lib.az_expand_wave.argtypes = [vp, vp, vp, vp, vp]
    lib.az_search_remaining.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp]
    lib.az_set_virtual_loss.argtypes = [vp, C.c_double]
    lib.az_draw_rng.argtypes = [vp, C.c_uint64, vp, C.c_float, vp, vp, vp]
    lib.az_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.az_tower_forward_h2.argtypes = [C.c_int32] * 4 + [vp] * 11 + [C.c_int32, vp, vp]
    lib.az_tower_forward_h1.argtypes = [C.c_int32] * 5 + [vp] * 11 + [C.c_int32, C.c_int32, vp, vp]
    lib.az_tower_last_error.argtypes = []
    lib.az_tower_last_error.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("az_last_error", "az_tower_last_error", "az_heads_feat_stride", "az_static_max_boards",
                        "az_smallnet_head_floats", "az_quick_stash_bytes"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def copy_to_host(dst: torch.Tensor, src: torch.Tensor, workgroups: int = 16) -> None:
    """src (device, contiguous) -> dst (pinned host, contiguous, same dtype and shape) on torch's current stream, by a few
    workgroups (az_copy_to_host): unlike a D2H ``copy_`` it leaves the chip to the kernels it overlaps with.  The data is
    valid once the stream has been synchronised (or an event recorded after the call has completed)."""
    if dst.dtype != src.dtype or dst.numel() != src.numel():
        raise EngineError("copy_to_host: dtype / size mismatch")
    if src.numel() == 0:
        return
    if not (dst.is_pinned() and dst.is_contiguous() and src.is_contiguous() and src.is_cuda):
        raise EngineError("copy_to_host: dst must be a contiguous pinned host tensor, src a contiguous device tensor")
    lib = load_library()
    with torch.cuda.device(src.device):
        st = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
        rc = lib.az_copy_to_host(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), src.numel() * src.element_size(), workgroups, st)
    if rc != 0:
        raise EngineError(f"az_copy_to_host failed ({rc})")


class EngineError(RuntimeError):
    pass


@dataclass
class Dims:
    n_action: int
    state_channels: int
    height: int
    width: int
    nn_channels: int
    max_plies: int
    state_words: int
    edge_cap: int
    node_cap: int
    depth_cap: int
    n_slots: int
    device_bytes: int

    @property
    def nn_numel(self) -> int:
        return self.nn_channels * self.height * self.width

    @property
    def state_numel(self) -> int:
        return self.state_channels * self.height * self.width


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """Many concurrent games, one search tree each, resident on one GPU."""

    def __init__(self, game: int, n_slots: int, max_sims: int, board_size: Optional[int] = None, c_puct: float = 1.0,
                 epsilon: float = 0.25, nn_dtype: torch.dtype = torch.float32, device=None, edge_cap: int = 0,
                 leaves_per_step: int = 1, virtual_loss: float = 0.3, win_length: int = 5, board_cols: int = 0,
                 history_len: int = 8):
        """leaves_per_step K > 1: virtual loss / in-tree leaf batching (select_wave / expand_wave); the per-simulation
        batch then has n_slots * K rows (``rows``).  Gomoku: board_size 3 .. 15 (default 15) and win_length 3 .. 6 are the
        reference's BOARD_SIZE / WIN_LENGTH constants (examples/gomoku/game.py:21-22).  Dots-and-boxes: board_size x board_cols =
        BOX_ROWS x BOX_COLS (examples/dots_and_boxes/game.py:24-25), 2 .. 5 each; default 3 x 3, board_cols 0 = square.
        history_len = HISTORY_LEN (config/basic.py:10), 1 .. 8: states in the dense network input and the emitted samples."""
        if board_size is None or (game != GAME_GOMOKU and board_size > 5):  # (older callers passed the gomoku default for every game)
            board_size = 15 if game == GAME_GOMOKU else 3
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the engine has no CPU path")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise EngineError(f"engine needs a cuda (HIP) device, got {self.device}")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        self.nn_dtype = nn_dtype
        cfg = AzConfig(game, board_size, n_slots, max_sims, c_puct, epsilon, _NN_DTYPES[nn_dtype], dev_index,
                       edge_cap, leaves_per_step, win_length, board_cols, history_len)
        self.win_length, self.history_len = win_length, history_len
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.az_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(f"az_create failed ({rc}): {self.lib.az_last_error(None).decode()}")
        self._h = h
        d = AzDims()
        self._ck(self.lib.az_get_dims(self._h, C.byref(d)))
        self.dims = Dims(d.n_action, d.state_channels, d.height, d.width, d.nn_channels, d.max_plies,
                         d.state_words, d.edge_cap, d.node_cap, d.depth_cap, d.n_slots, d.device_bytes)
        self.game = game
        self.n_slots = n_slots
        self.max_sims = max_sims
        self.leaves_per_step = max(int(d.leaves_per_step), 1)
        self.rows = n_slots * self.leaves_per_step  # rows of the NN batch / logits / values / packed leaf batch
        if self.leaves_per_step > 1:
            self._ck(self.lib.az_set_virtual_loss(self._h, C.c_double(virtual_loss)))

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc != 0:
            raise EngineError(f"az call failed ({rc}): {self.lib.az_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _expect(self, t, shape, dtype, name):
        if t is None:
            return
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            raise EngineError(f"{name}: expected contiguous {dtype} {tuple(shape)} on {self.device}, got "
                              f"{t.dtype} {tuple(t.shape)} on {t.device} (contiguous={t.is_contiguous()})")

    def close(self):
        if getattr(self, "_h", None):
            self._flags_view = None  # (aliases engine memory: the caller must not use a view obtained earlier after close())
            self.lib.az_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def new_nn_input(self):
        d = self.dims
        return torch.zeros(self.rows, d.nn_channels, d.height, d.width, dtype=self.nn_dtype, device=self.device)

    # ------------------------------------------------------------------ entry points
    def select_wave(self, sims_target: int, nn_input=None):
        """up to K descents per tree with virtual loss (az_select_wave); leaf j of slot g is row g*K + j"""
        d = self.dims
        self._expect(nn_input, (self.rows, d.nn_channels, d.height, d.width), self.nn_dtype, "nn_input")
        self._ck(self.lib.az_select_wave(self._h, _ptr(nn_input), int(sims_target), self._stream()))

    def expand_wave(self, logits, values, noise=None):
        A = self.dims.n_action
        self._expect(logits, (self.rows, A), torch.float32, "logits")
        if values.dim() == 2:
            values = values.view(-1)
        self._expect(values, (self.rows,), torch.float32, "values")
        self._expect(noise, (self.n_slots, A), torch.float32, "noise")
        self._ck(self.lib.az_expand_wave(self._h, _ptr(logits), _ptr(values), _ptr(noise), self._stream()))

    def search_remaining(self, sims_target: int):
        """-> (simulations left over all active trees, the most any one tree has left); synchronises"""
        tot, mx = C.c_int64(0), C.c_int32(0)
        self._ck(self.lib.az_search_remaining(self._h, int(sims_target), C.byref(tot), C.byref(mx), self._stream()))
        return int(tot.value), int(mx.value)

    def check(self):
        self._ck(self.lib.az_check(self._h, self._stream()))

    def reset_games(self, mask=None):
        self._expect(mask, (self.n_slots,), torch.uint8, "mask")
        self._ck(self.lib.az_reset_games(self._h, _ptr(mask), self._stream()))

    def set_position(self, slot: int, history, root_player: int):
        h = torch.as_tensor(history, dtype=torch.float32, device="cpu").contiguous()
        if h.numel() != self.history_len * self.dims.state_numel:
            raise EngineError(f"history must hold {self.history_len} states of {self.dims.state_numel} values")
        self._ck(self.lib.az_set_position(self._h, slot, C.c_void_p(h.data_ptr()), int(root_player), self._stream()))

    def begin_search(self):
        self._ck(self.lib.az_begin_search(self._h, self._stream()))

    def leaf_buffers(self):
        """device addresses (ints) of the packed leaf batch: (ring u32 (G,8,SW), leaf records int32 (G,4),
        list of slots to evaluate i32 (G), their count i32, per-slot needs-the-network flags u8 (G))"""
        ring, leaf, lst, cnt, flg = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self.lib.az_leaf_buffers(self._h, C.byref(ring), C.byref(leaf), C.byref(lst), C.byref(cnt), C.byref(flg)))
        return ring.value, leaf.value, lst.value, cnt.value, flg.value

    def leaf_flags(self) -> torch.Tensor:
        """the per-row needs-the-network flags of the last select_gather as a uint8 tensor (rows,) that ALIASES the engine's
        buffer (zero copy, through __cuda_array_interface__): 1 = the row's leaf wants an evaluation, 0 = terminal / inactive.
        Valid until close()."""
        if getattr(self, "_flags_view", None) is None:
            ptr, rows = self.leaf_buffers()[4], self.rows

            class _Dev:
                __cuda_array_interface__ = {"shape": (rows,), "typestr": "|u1", "data": (ptr, False), "version": 2}

            with torch.cuda.device(self.device):
                self._flags_view = torch.as_tensor(_Dev(), device=self.device)
        return self._flags_view

    def compact_leaves(self):
        """fill the list/count of leaf_buffers() for the batch of the last select_gather"""
        self._ck(self.lib.az_compact_leaves(self._h, self._stream()))

    def select_gather(self, nn_input=None):
        """nn_input None: only the packed leaf batch is produced (what the fused tower reads)"""
        d = self.dims
        self._expect(nn_input, (self.rows, d.nn_channels, d.height, d.width), self.nn_dtype, "nn_input")
        self._ck(self.lib.az_select_gather(self._h, _ptr(nn_input), self._stream()))

    def expand_backup(self, logits, values, noise=None):
        A = self.dims.n_action
        self._expect(logits, (self.n_slots, A), torch.float32, "logits")
        if values.dim() == 2:
            values = values.view(-1)
        self._expect(values, (self.n_slots,), torch.float32, "values")
        self._expect(noise, (self.n_slots, A), torch.float32, "noise")
        self._ck(self.lib.az_expand_backup(self._h, _ptr(logits), _ptr(values), _ptr(noise), self._stream()))

    def expand_select(self, logits, values, noise=None, nn_input=None):
        """expand_backup of the running simulation + select_gather of the next one, in one launch"""
        d = self.dims
        self._expect(logits, (self.n_slots, d.n_action), torch.float32, "logits")
        if values.dim() == 2:
            values = values.view(-1)
        self._expect(values, (self.n_slots,), torch.float32, "values")
        self._expect(noise, (self.n_slots, d.n_action), torch.float32, "noise")
        self._expect(nn_input, (self.n_slots, d.nn_channels, d.height, d.width), self.nn_dtype, "nn_input")
        self._ck(self.lib.az_expand_select(self._h, _ptr(logits), _ptr(values), _ptr(noise), _ptr(nn_input), self._stream()))

    def root_policy(self, pi=None, visits=None, child_q=None, root_prior=None):
        A, G = self.dims.n_action, self.n_slots
        pi = self.new(G, A) if pi is None else pi
        self._expect(pi, (G, A), torch.float32, "pi")
        self._expect(visits, (G, A), torch.int32, "visits")
        self._expect(child_q, (G, A), torch.float64, "child_q")
        self._expect(root_prior, (G, A), torch.float32, "root_prior")
        self._ck(self.lib.az_root_policy(self._h, _ptr(pi), _ptr(visits), _ptr(child_q), _ptr(root_prior),
                                         self._stream()))
        return pi

    def root_nn_state(self, out=None):
        d = self.dims
        out = torch.zeros(self.n_slots, d.nn_channels, d.height, d.width, device=self.device) if out is None else out
        self._expect(out, (self.n_slots, d.nn_channels, d.height, d.width), torch.float32, "nn_state")
        self._ck(self.lib.az_root_nn_state(self._h, _ptr(out), self._stream()))
        return out

    def play_moves(self, inv_tao, expo, forced=None, actions=None, done=None):
        A, G = self.dims.n_action, self.n_slots
        actions = self.new(G, dtype=torch.int32) if actions is None else actions
        done = self.new(G, dtype=torch.uint8) if done is None else done
        self._expect(inv_tao, (G,), torch.float32, "inv_tao")
        self._expect(expo, (G, A), torch.float32, "expo")
        self._expect(forced, (G,), torch.int32, "forced")
        self._expect(actions, (G,), torch.int32, "actions")
        self._expect(done, (G,), torch.uint8, "done")
        self._ck(self.lib.az_play_moves(self._h, _ptr(inv_tao), _ptr(expo), _ptr(forced), _ptr(actions), _ptr(done),
                                        self._stream()))
        return actions, done

    def draw_rng(self, seed: int, game_ids, noise=None, expo=None, alpha: float = 0.3):
        """per-game counter-based rows (az_draw_rng): Dirichlet(alpha) into ``noise`` and / or Exponential(1) into ``expo``
        for every active slot, keyed on (seed, game_ids[slot], the slot's ply)"""
        A, G = self.dims.n_action, self.n_slots
        self._expect(game_ids, (G,), torch.int64, "game_ids")
        self._expect(noise, (G, A), torch.float32, "noise")
        self._expect(expo, (G, A), torch.float32, "expo")
        self._ck(self.lib.az_draw_rng(self._h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _ptr(game_ids), C.c_float(alpha),
                                      _ptr(noise), _ptr(expo), self._stream()))

    def game_info(self):
        G = self.n_slots
        plies, player = self.new(G, dtype=torch.int32), self.new(G, dtype=torch.int32)
        active, result = self.new(G, dtype=torch.uint8), self.new(G)
        self._ck(self.lib.az_game_info(self._h, _ptr(plies), _ptr(player), _ptr(active), _ptr(result), self._stream()))
        return plies, player, active, result

    def emit_samples(self, slots, plies):
        """slots: finished slot ids (list / 1-D int tensor, host); plies: their lengths.
        -> (states (n, C_in*H*W), pis (n, A), z (n,), offsets) in the reference's put order per game."""
        d = self.dims
        slots_h = torch.as_tensor(slots, dtype=torch.int32, device="cpu").reshape(-1)
        plies_h = torch.as_tensor(plies, dtype=torch.int32, device="cpu").reshape(-1)
        n = int(slots_h.numel())
        offs = torch.zeros(n + 1, dtype=torch.int32)
        if n:
            offs[1:] = torch.cumsum(plies_h, 0)
        total = int(offs[-1])
        states = self.new(total, d.nn_numel)
        pis = self.new(total, d.n_action)
        z = self.new(total)
        if n:
            s_dev = slots_h.to(self.device)
            o_dev = offs.to(self.device)
            self._ck(self.lib.az_emit_samples(self._h, _ptr(s_dev), _ptr(o_dev), n, _ptr(states), _ptr(pis), _ptr(z),
                                              self._stream()))
        return states, pis, z, offs

    def augment_samples(self, states, pis, zs, out=None, ring_capacity: int = 0, ring_write_idx: int = 0):
        """8-fold symmetry expansion on the device (the pools' get_enhanced_data); -> (states (8n,.), pis, z) or
        writes rows (ring_write_idx + 8 i + t) % ring_capacity of the given out buffers."""
        d = self.dims
        n = int(states.shape[0])
        states = states.reshape(n, d.nn_numel)
        self._expect(states, (n, d.nn_numel), torch.float32, "states")
        self._expect(pis, (n, d.n_action), torch.float32, "pis")
        self._expect(zs, (n,), torch.float32, "zs")
        if out is None:
            out = (self.new(8 * n, d.nn_numel), self.new(8 * n, d.n_action), self.new(8 * n))
        os_, op, oz = out
        rows = ring_capacity or 8 * n
        if os_.numel() < rows * d.nn_numel or op.numel() < rows * d.n_action or oz.numel() < rows:
            raise EngineError("augment_samples: out buffers are smaller than the ring / 8n rows")
        self._ck(self.lib.az_augment_samples(self._h, n, _ptr(states), _ptr(pis), _ptr(zs), _ptr(os_), _ptr(op), _ptr(oz),
                                             ring_capacity, ring_write_idx, self._stream()))
        return out

    def read_stats(self) -> dict:
        s = AzStats()
        self._ck(self.lib.az_read_stats(self._h, C.byref(s), self._stream()))
        return {k: int(getattr(s, k)) for k, _ in AzStats._fields_}

    def reset_stats(self):
        self._ck(self.lib.az_reset_stats(self._h, self._stream()))

    def eval_hash(self, nn_input, logits, values, scale: float = 4.0):
        self._ck(self.lib.az_eval_hash(self._h, _ptr(nn_input), C.c_float(scale), _ptr(logits), _ptr(values),
                                       self._stream()))
