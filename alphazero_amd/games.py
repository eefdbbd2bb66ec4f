"""Host-side mirror of the reference's game contract (games/base.py:9-150) for the three games that
have native rule kernels.  Same static interface on CPU float32 tensors --

    initial_state() / next_state(state, action, player) / legal_action_mask(state) /
    is_terminal(state) / _get_winner(state) / terminal_evaluation(state, player) /
    get_player_channel(state, player)

-- and the same instance surface (reset / step / get_state / get_player / get_history / get_legal_mask
/ evaluation), so code written against the reference's ``BaseGame`` (UI, evaluation scripts, the
``MCTS.search`` adapter's callers) runs against these classes on a machine where the reference is
absent.  The GPU engine never calls them: rules run as bitboard kernels (csrc/az_games.hip.h); these
exist for positions entering and leaving the engine and for tests.  PLAYER1 = 1, PLAYER2 = -1
(config/basic.py:3-4); rule violations raise ValueError like the reference.
"""
from __future__ import annotations

from collections import deque

import torch

from .engine import GAME_DAB, GAME_GOMOKU, GAME_TICTACTOE

PLAYER1, PLAYER2 = 1, -1
HISTORY_LEN = 8


class BaseGame:
    AZ_GAME: tuple  # (engine game id, board size)
    HISTORY_LEN = HISTORY_LEN  # config/basic.py:10 (a module constant in the reference; per class here: with_history_len)

    def __init__(self) -> None:
        self.reset()

    # ---- instance surface (games/base.py:15-50)
    def reset(self) -> None:
        self._state = self.initial_state()
        self._player = PLAYER1
        self._history = deque([self.initial_state() for _ in range(self.HISTORY_LEN)], maxlen=self.HISTORY_LEN)

    def step(self, action: int) -> bool:
        self._state, self._player = self.next_state(self._state, action, self._player)
        self._history.append(self._state.clone())
        return self.is_terminal(self._state)

    def get_player(self) -> int:
        return self._player

    def get_state(self) -> torch.Tensor:
        return self._state.detach().clone()

    def get_history(self) -> torch.Tensor:
        """(HISTORY_LEN,H,W) for 2-D states, (HISTORY_LEN*C,H,W) for 3-D ones, oldest first (utils/history_manager.py:41-47)"""
        states = list(self._history)
        return (torch.stack(states, 0) if states[0].dim() == 2 else torch.cat(states, 0)).clone()

    def get_legal_mask(self) -> torch.Tensor:
        return self.legal_action_mask(self._state)

    def evaluation(self) -> float:
        return self.terminal_evaluation(self._state, self._player)

    # ---- static contract
    @classmethod
    def terminal_evaluation(cls, state: torch.Tensor, player: int) -> float:
        w = cls._get_winner(state)
        return 1.0 if w == player else (-1.0 if w == -player else 0.0)

    @staticmethod
    def get_player_channel(state: torch.Tensor, player: int) -> torch.Tensor:
        return torch.full((1, *state.shape[-2:]), player)


def with_history_len(game_cls, history_len: int):
    """game_cls with the reference's HISTORY_LEN constant (config/basic.py:10) set to 1 .. 8: the engine then builds network
    inputs and samples of history_len * C + 1 planes"""
    h = int(history_len)
    if not 1 <= h <= 8:
        raise ValueError("with_history_len: 1 .. 8")
    return game_cls if h == game_cls.HISTORY_LEN else type(f"{game_cls.__name__}H{h}", (game_cls,), {"HISTORY_LEN": h})


class _StoneGame(BaseGame):
    """boards where a move puts +-1 on an empty cell and players alternate"""
    N = 3

    @classmethod
    def initial_state(cls) -> torch.Tensor:
        return torch.zeros((cls.N, cls.N), dtype=torch.float32)

    @classmethod
    def next_state(cls, state: torch.Tensor, action: int, player: int):
        if not 0 <= action < cls.N * cls.N:
            raise ValueError("action out of range")
        if state.view(-1)[action].item() != 0:
            raise ValueError("illegal move: cell occupied")
        nxt = state.clone()
        nxt.view(-1)[action] = float(player)
        return nxt, -player

    @staticmethod
    def legal_action_mask(state: torch.Tensor) -> torch.Tensor:
        return (state.reshape(1, -1) == 0).float()

    @classmethod
    def is_terminal(cls, state: torch.Tensor) -> bool:
        return cls._get_winner(state) is not None or int(torch.count_nonzero(state)) == cls.N * cls.N


class TicTacToeGame(_StoneGame):
    """examples/tictactoe/game.py"""
    N = 3
    AZ_GAME = (GAME_TICTACTOE, 3)

    @staticmethod
    def _get_winner(state: torch.Tensor):
        sums = torch.cat([state.sum(1), state.sum(0), torch.diag(state).sum().view(1),
                          torch.diag(torch.fliplr(state)).sum().view(1)])
        for s in sums.tolist():  # rows, columns, diagonal, anti-diagonal: first complete line wins
            if s == 3:
                return PLAYER1
            if s == -3:
                return PLAYER2
        return None


class GomokuGame(_StoneGame):
    """examples/gomoku/game.py (five or more in a row wins; BOARD_SIZE is a class attribute)"""
    BOARD_SIZE = 15
    WIN_LENGTH = 5
    N = 15
    AZ_GAME = (GAME_GOMOKU, 15)

    @classmethod
    def _get_winner(cls, state: torch.Tensor):
        # A position reached by legal play holds fives of at most one colour, so "which colour has a
        # five" equals the reference's row-major first-hit scan (game.py:62-89).
        n, k = cls.N, cls.WIN_LENGTH
        for colour in (PLAYER1, PLAYER2):
            m = (state == colour).float()
            runs = [m.unfold(1, k, 1).sum(-1), m.unfold(0, k, 1).sum(-1)]
            d = torch.stack([m[i:n - k + 1 + i, i:n - k + 1 + i] for i in range(k)]).sum(0)
            a = torch.stack([m[i:n - k + 1 + i, k - 1 - i:n - i] for i in range(k)]).sum(0)
            if any(bool((r == k).any()) for r in runs + [d, a]):
                return colour
        return None


class Gomoku9Game(GomokuGame):
    BOARD_SIZE = 9
    N = 9
    AZ_GAME = (GAME_GOMOKU, 9)


def gomoku_game(board_size: int, win_length: int = 5):
    """GomokuGame with the reference's two constants set (examples/gomoku/game.py:21-22 BOARD_SIZE, WIN_LENGTH): any board
    3 .. 15 with 3 .. 6 in a row to win has a native rule kernel (az::Gomoku<N> with the win length as a run-time value)."""
    n, k = int(board_size), int(win_length)
    if not (3 <= n <= 15 and 3 <= k <= 6 and k <= n):
        raise ValueError("gomoku_game: board_size 3 .. 15, win_length 3 .. 6 and <= board_size")
    if (n, k) == (15, 5):
        return GomokuGame
    if (n, k) == (9, 5):
        return Gomoku9Game
    return type(f"Gomoku{n}x{n}k{k}Game", (GomokuGame,), {"BOARD_SIZE": n, "N": n, "WIN_LENGTH": k, "AZ_GAME": (GAME_GOMOKU, n)})


class DotsAndBoxesGame(BaseGame):
    """examples/dots_and_boxes/game.py: BOX_ROWS x BOX_COLS boxes (3 x 3 as shipped); state (3, BOX_ROWS + 1, BOX_COLS + 1) =
    horizontal edges, vertical edges, box owners; action a < H_EDGE_COUNT: horizontal edge (a // BOX_COLS, a % BOX_COLS), else
    vertical edge (idx // DOT_COLS, idx % DOT_COLS); completing a box keeps the turn."""
    BOX_ROWS = BOX_COLS = 3
    NUM_ACTION = 24
    AZ_GAME = (GAME_DAB, 3)

    @classmethod
    def initial_state(cls) -> torch.Tensor:
        return torch.zeros((3, cls.BOX_ROWS + 1, cls.BOX_COLS + 1), dtype=torch.float32)

    @classmethod
    def _edge(cls, action: int):
        nh, bc = (cls.BOX_ROWS + 1) * cls.BOX_COLS, cls.BOX_COLS
        return (0, action // bc, action % bc) if action < nh else (1, (action - nh) // (bc + 1), (action - nh) % (bc + 1))

    @classmethod
    def next_state(cls, state: torch.Tensor, action: int, player: int):
        if not 0 <= action < cls.NUM_ACTION:
            raise ValueError("action out of range")
        ch, r, c = cls._edge(action)
        if state[ch, r, c].item() != 0:
            raise ValueError("illegal move: edge already drawn")
        nxt = state.clone()
        nxt[ch, r, c] = 1.0
        boxes = [(r - 1, c), (r, c)] if ch == 0 else [(r, c - 1), (r, c)]
        took = False
        for br, bc in boxes:
            if 0 <= br < cls.BOX_ROWS and 0 <= bc < cls.BOX_COLS and nxt[2, br, bc].item() == 0 and \
                    min(nxt[0, br, bc], nxt[0, br + 1, bc], nxt[1, br, bc], nxt[1, br, bc + 1]).item() != 0:
                nxt[2, br, bc] = float(player)
                took = True
        return nxt, (player if took else -player)

    @classmethod
    def legal_action_mask(cls, state: torch.Tensor) -> torch.Tensor:
        r, c = cls.BOX_ROWS, cls.BOX_COLS
        return torch.cat([(state[0, :r + 1, :c] == 0).reshape(-1), (state[1, :r, :c + 1] == 0).reshape(-1)]).float().view(1, cls.NUM_ACTION)

    @classmethod
    def is_terminal(cls, state: torch.Tensor) -> bool:
        r, c = cls.BOX_ROWS, cls.BOX_COLS
        return int(state[0, :r + 1, :c].sum() + state[1, :r, :c + 1].sum()) == cls.NUM_ACTION

    @classmethod
    def _get_winner(cls, state: torch.Tensor):
        if not cls.is_terminal(state):
            return None
        own = state[2, :cls.BOX_ROWS, :cls.BOX_COLS]
        p1, p2 = int((own == PLAYER1).sum()), int((own == PLAYER2).sum())
        return PLAYER1 if p1 > p2 else (PLAYER2 if p2 > p1 else None)


def dots_and_boxes_game(box_rows: int, box_cols: int = 0):
    """DotsAndBoxesGame with the reference's two constants set (examples/dots_and_boxes/game.py:24-25 BOX_ROWS, BOX_COLS): any
    board of 2 .. 5 by 2 .. 5 boxes has a native rule kernel (az::Dab<R, C>)."""
    r = int(box_rows)
    c = int(box_cols) or r
    if not (2 <= r <= 5 and 2 <= c <= 5):
        raise ValueError("dots_and_boxes_game: box_rows and box_cols 2 .. 5")
    if (r, c) == (3, 3):
        return DotsAndBoxesGame
    return type(f"DotsAndBoxes{r}x{c}Game", (DotsAndBoxesGame,),
                {"BOX_ROWS": r, "BOX_COLS": c, "NUM_ACTION": (r + 1) * c + r * (c + 1), "AZ_GAME": (GAME_DAB, r)})
