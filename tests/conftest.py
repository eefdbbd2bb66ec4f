import os
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
        return 1, 3
    if tag.startswith("gomoku15"):
        return 2, 15
    if tag.startswith("gomoku9"):
        return 2, 9
    raise KeyError(tag)
