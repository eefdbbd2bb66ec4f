"""numpy restatement of the 8-fold sample augmentation the reference's pools apply on put.

TEST INFRASTRUCTURE ONLY (see oracle/az_oracle.c header).  Pinned by tests/golden/augment.npz, generated
from the reference (tests/golden/make_golden.py::make_augment).

  utils/data_enhancer.py:13-70                 stone games: for k in 0..3: rot90(k) then rot90(k)+flip(last dim)
  examples/dots_and_boxes/game.py:158-320      dots-and-boxes: edges/boxes permuted; "flip" is a transpose
Order of the 8 outputs per sample: (k=0), (k=0,flip), (k=1), (k=1,flip), ...
"""
import numpy as np


def augment_stone(state, pi, z):
    """state (C,H,W), pi (H*W,) -> lists of 8 states / pis / zs (data_enhancer.py:29-39)"""
    h, w = state.shape[-2:]
    outs, outp = [], []
    for k in range(4):
        rs = np.rot90(state, k=k, axes=(-2, -1))                      # DataEnhancer.rotate :48-57
        rp = np.rot90(pi.reshape(h, w), k=k, axes=(0, 1)).reshape(-1)
        outs.append(rs.copy()); outp.append(rp.copy())
        outs.append(rs[..., ::-1].copy())                              # DataEnhancer.flip :59-70
        outp.append(rp.reshape(h, w)[:, ::-1].reshape(-1).copy())
    return outs, outp, [z] * 8


# ---- dots and boxes (B x B boxes, D x D dots, D = B + 1; 3 x 3 boxes in the reference as shipped -- the symmetries exist
# for square boards only: game.py:318-319 raises otherwise)
def _rotate_point(pt, k, rows, cols):  # game.py:305-313
    r, c = pt
    for _ in range(k % 4):
        r, c = cols - 1 - c, r
        rows, cols = cols, rows
    return (r, c), rows, cols


def _transform_edge(etype, r, c, k, flip, B=3):  # game.py:261-292
    p1 = (r, c)
    p2 = (r, c + 1) if etype == "H" else (r + 1, c)
    p1, _, _ = _rotate_point(p1, k, B + 1, B + 1)
    p2, _, _ = _rotate_point(p2, k, B + 1, B + 1)
    if flip:  # _flip_point: transpose
        p1, p2 = (p1[1], p1[0]), (p2[1], p2[0])
    if p1[0] == p2[0]:
        return "H", p1[0], min(p1[1], p2[1])
    return "V", min(p1[0], p2[0]), p1[1]


def _transform_box(r, c, k, flip, B=3):  # game.py:294-303
    (r, c), _, _ = _rotate_point((r, c), k, B, B)
    return (c, r) if flip else (r, c)


def _a2e(a, B=3):  # game.py:88-97
    nh = (B + 1) * B
    return ("H", a // B, a % B) if a < nh else ("V", (a - nh) // (B + 1), (a - nh) % (B + 1))


def _e2a(t, r, c, B=3):  # game.py:99-103
    return r * B + c if t == "H" else (B + 1) * B + r * (B + 1) + c


def dab_edge_perm(k, flip, B=3):
    """perm[a] = new action of edge a"""
    return [_e2a(*_transform_edge(*_a2e(a, B), k, flip, B), B) for a in range(2 * B * (B + 1))]


def dab_box_perm(k, flip, B=3):
    out = []
    for b in range(B * B):
        r, c = _transform_box(b // B, b % B, k, flip, B)
        out.append(r * B + c)
    return out


def augment_dab(state, pi, z):
    """state (3 HISTORY_LEN + 1,D,D), pi (2 B D,) (game.py:158-259)"""
    B = state.shape[-1] - 1
    if state.shape[-2] != state.shape[-1]:
        raise ValueError("the dots-and-boxes symmetries need a square board (game.py:318-319)")
    A = 2 * B * (B + 1)
    outs, outp = [], []
    for k in range(4):
        for flip in (False, True):
            ep, bp = dab_edge_perm(k, flip, B), dab_box_perm(k, flip, B)
            o = np.zeros_like(state)
            for blk in range((state.shape[0] - 1) // 3):  # HISTORY_LEN blocks of 3 planes (game.py:182-203)
                base = 3 * blk
                for a in range(A):
                    t, r, c = _a2e(a, B)
                    v = state[base + (0 if t == "H" else 1), r, c]
                    if v != 0:
                        nt, nr, nc = _a2e(ep[a], B)
                        o[base + (0 if nt == "H" else 1), nr, nc] = v
                for b in range(B * B):
                    v = state[base + 2, b // B, b % B]
                    if v != 0:
                        o[base + 2, bp[b] // B, bp[b] % B] = v
            o[-1] = state[-1]
            p = np.zeros_like(pi)
            for a in range(A):
                p[ep[a]] = pi[a]
            outs.append(o); outp.append(p)
    return outs, outp, [z] * 8


def augment(game, state, pi, z):
    return augment_dab(state, pi, z) if game == 1 else augment_stone(state, pi, z)
