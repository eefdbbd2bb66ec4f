import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/), built on demand with gcc."""
    from oracle import az_oracle

    az_oracle.build()
    return az_oracle


def load_cases(npz_name):
    import numpy as np

    z = np.load(os.path.join(GOLDEN, npz_name))
    cases = {}
    for tag in z["tags"]:
        tag = str(tag)
        cases[tag] = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(tag + "__")}
    return cases


def game_of(tag):
    """fixture tag -> (game id, board size)"""
    if tag.startswith("tictactoe"):
        return 0, 3
    if tag.startswith("dab"):
        return 1, box_of(tag)[0]
    m = re.match(r"gomoku(\d+)", tag)
    if m:
        return 2, int(m.group(1))
    raise KeyError(tag)


def win_of(tag):
    """fixture tag -> the reference's WIN_LENGTH the fixture was recorded under ("gomoku7k4_1" -> 4; 5 where the tag does not say)"""
    m = re.match(r"gomoku\d+k(\d)", tag)
    return int(m.group(1)) if m else 5


def box_of(tag):
    """fixture tag -> the reference's (BOX_ROWS, BOX_COLS) the fixture was recorded under ("dab4x5_0" -> (4, 5); 3 x 3 where the tag does
    not say)"""
    m = re.match(r"dab(\d)x(\d)", tag)
    return (int(m.group(1)), int(m.group(2))) if m else (3, 3)


def hist_of(tag):
    """fixture tag -> the reference's HISTORY_LEN the fixture was recorded under ("dab_h2_0" -> 2; 8 where the tag does not say)"""
    m = re.search(r"_h(\d)(_|$)", tag)
    return int(m.group(1)) if m else 8


def cols_of(tag):
    """the board_cols argument of Engine for a fixture tag (BOX_COLS of a dots-and-boxes fixture, 0 otherwise)"""
    return box_of(tag)[1] if tag.startswith("dab") else 0


class reference_constants:
    """``with reference_constants(oracle, tag):`` the oracle under the constants the reference had when the fixture was recorded:
    WIN_LENGTH of gomoku<N>k<K>, BOX_ROWS x BOX_COLS of dab<R>x<C>, HISTORY_LEN of <game>_h<H>"""

    def __init__(self, oracle, tag):
        self.cm = [oracle.win_length(win_of(tag)), oracle.box_shape(*box_of(tag)), oracle.history_len(hist_of(tag))]

    def __enter__(self):
        for c in self.cm:
            c.__enter__()
        return self

    def __exit__(self, *exc):
        for c in reversed(self.cm):
            c.__exit__(*exc)
        return False


# gomoku under other values of the reference's BOARD_SIZE / WIN_LENGTH constants (tests/golden/make_golden.py mnk)
MNK_RULES = [(f"gomoku{n}k{k}", 2, n) for n, k in ((7, 4), (11, 5), (6, 3), (13, 6), (5, 5), (3, 3), (12, 4), (8, 6))]
# dots-and-boxes under other values of the reference's BOX_ROWS / BOX_COLS constants (make_golden.py dabshapes)
DAB_SHAPES = [(2, 2), (4, 4), (5, 5), (2, 3), (3, 2), (4, 5), (5, 4), (3, 5)]
DAB_RULES = [(f"dab{r}x{c}", 1, r) for r, c in DAB_SHAPES]
