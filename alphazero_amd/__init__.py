"""alphazero_amd -- MI355X-native batched MCTS self-play engine (drop-in for the self-play path of
BlueRose2020/alphazero: core/MCTS_alphazero.py + training/self_play.py)."""
from .engine import (Engine, EngineError, GAME_DAB, GAME_GOMOKU, GAME_TICTACTOE, load_library)  # noqa: F401

__all__ = ["Engine", "EngineError", "GAME_TICTACTOE", "GAME_DAB", "GAME_GOMOKU", "load_library"]
