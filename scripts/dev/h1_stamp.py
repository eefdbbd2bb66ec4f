"""Where a workgroup's time goes in the row-tile tower (az_tower_h1.hip), from in-kernel s_memtime stamps of every workgroup's
10th board.  Needs a diagnostic build:  scripts/dev/build_variant.sh h1stamp -DAZ_H1_STAMP=1
   AZ_ENGINE_LIB=$PWD/alphazero_amd/lib/dev/h1stamp.so python scripts/dev/h1_stamp.py [bf16r|f16]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from alphazero_amd.engine import load_library
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    from alphazero_amd import Engine

    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16r"
    B = 16384
    torch.manual_seed(0)
    net = GomokuNet(15, 64, 2).eval().cuda()
    ev = FusedTowerEvaluator(net, prec)
    # the engine's packed leaf batch after a few plies (the path the bench times), every slot flagged
    eng = Engine(2, B, 8, board_size=15, nn_dtype=torch.float32, device="cuda:0")
    A = eng.dims.n_action
    lg, vl = eng.new(B, A), eng.new(B)
    it = torch.ones(B, device="cuda")
    for _ in range(8):
        eng.begin_search()
        for _ in range(3):
            eng.select_gather(None)
            eng.eval_hash(None, lg, vl, 6.0)
            eng.expand_backup(lg, vl, None)
        eng.play_moves(it, torch.empty(B, A, device="cuda").exponential_(1))
    eng.begin_search()
    eng.select_gather(None)
    ring, leaf, _, _, flags = eng.leaf_buffers()
    for _ in range(20):
        ev(None, logits_out=lg, packed=(ring, leaf, None, None, flags), n_boards=B)
    torch.cuda.synchronize()
    lib = load_library()
    buf = np.zeros(512 * 4 * 32, dtype=np.uint64)
    lib.az_h1_stamps.argtypes = [C.c_void_p]
    assert lib.az_h1_stamps(buf.ctypes.data) == 0
    t = buf.reshape(512, 4, 32).astype(np.int64)
    ok = t[:, :, 0] > 0
    print(f"{int(ok.sum())} waves stamped")
    names = {0: "top", 1: "input+barrier", 2: "stem conv", 3: "barrier", 4: "epilogue+ticket", 5: "barrier"}
    for l in range(4):
        names[6 + 4 * l] = f"conv {l}"
        names[7 + 4 * l] = "barrier"
        names[8 + 4 * l] = "epilogue" + (" (res)" if l % 2 == 0 else "")
        names[9 + 4 * l] = "barrier"
    names[30] = "heads"
    names[31] = "park"
    order = list(range(0, 22)) + [30, 31]
    # tile count of a wave: chalf = (wave ^ (block >> 8)) & 1 -> 8 tiles if 0, 7 if 1
    blk = np.arange(512)[:, None]
    wv = np.arange(4)[None, :]
    chalf = (wv ^ (blk >> 8)) & 1
    for label, sel in (("8-tile waves", ok & (chalf == 0)), ("7-tile waves", ok & (chalf == 1))):
        print(f"-- {label}: mean cycles per phase (whole board {np.mean((t[:, :, 31] - t[:, :, 0])[sel]):.0f})")
        prev = 0
        for k in order[1:]:
            d = (t[:, :, k] - t[:, :, prev])[sel]
            print(f"   {names[k]:>18s}: {d.mean():8.0f}  (p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f})")
            prev = k
