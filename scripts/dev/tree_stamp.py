"""When does every wave of k_expand_select start and end?  (-DAZ_TREE_STAMP=1 build: scripts/dev/build_variant.sh treestamp -DAZ_TREE_STAMP=1)
   AZ_ENGINE_LIB=$PWD/alphazero_amd/lib/dev/treestamp.so python scripts/dev/tree_stamp.py [slots] [sims]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from alphazero_amd import Engine
    from alphazero_amd.engine import load_library

    G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    lib = load_library()
    eng = Engine(2, G, 400, board_size=15, nn_dtype=torch.bfloat16, device="cuda:0")
    A = eng.dims.n_action
    lg, vl = eng.new(G, A), eng.new(G)
    noise = torch._sample_dirichlet(torch.full((G, A), 0.3, device="cuda"))
    eng.begin_search()
    eng.select_gather(None)
    for _ in range(sims):
        eng.eval_hash(None, lg, vl, 6.0)
        eng.expand_select(lg, vl, noise, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.eval_hash(None, lg, vl, 6.0)
    e0.record(); eng.expand_select(lg, vl, noise, None); e1.record()
    torch.cuda.synchronize()
    n = min(G, 65536)
    buf = np.zeros((n, 3), np.uint64)
    assert lib.az_tree_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
    t0 = buf[:, 0].min()
    st, en = (buf[:, 0] - t0).astype(np.float64) / 100.0, (buf[:, 1] - t0).astype(np.float64) / 100.0  # us
    life = en - st
    print(f"{G} trees, launch {e0.elapsed_time(e1) * 1e3:.1f} us by events; first start -> last end {en.max():.1f} us")
    print(f"wave lifetime: mean {life.mean():.2f} us, p10 {np.percentile(life, 10):.2f}, p50 {np.percentile(life, 50):.2f}, p90 {np.percentile(life, 90):.2f}, max {life.max():.2f}")
    print(f"last wave STARTS at {st.max():.1f} us; starts per us over the dispatch: {n / max(st.max(), 1e-9):.0f}")
    edges = np.arange(0, en.max() + 2, 2.0)
    res = [(int(((st <= t) & (en > t)).sum())) for t in edges]
    print("resident waves every 2 us:", res)
    hw = buf[:, 2]
    cu = ((hw >> np.uint64(32)) << np.uint64(16)) | (((hw >> np.uint64(13)) & np.uint64(7)) << np.uint64(8)) | ((hw >> np.uint64(8)) & np.uint64(15))
    ids, cnt = np.unique(cu, return_counts=True)
    print(f"{len(ids)} distinct (xcc, se, cu) ids; waves per id: min {cnt.min()} mean {cnt.mean():.1f} max {cnt.max()}")
    eng.close()
