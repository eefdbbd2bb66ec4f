"""Print the per-phase cycle stamps of the -DAZ_F32_STAMP=1 build of the f32 tower (alphazero_amd/lib/dev/f32stamp.so):
   bash scripts/dev/build_variant.sh f32stamp -DAZ_F32_STAMP=1 && python scripts/dev/f32_stamp.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alphazero_amd.fused import FusedTowerEvaluator
from alphazero_amd.nets import GomokuNet

if __name__ == "__main__":
    torch.manual_seed(0)
    N = int(os.environ.get("BOARD", "15"))  # (TIME mode: BOARD=9 BATCH=4096 for BASELINE config 3's tower)
    B = int(os.environ.get("BATCH", "16384"))
    net = GomokuNet(N, 64, 2).eval().cuda()
    ev = FusedTowerEvaluator(net, "f32")
    x = torch.randint(-1, 2, (B, 9, N, N), device="cuda").float()
    ev(x)
    feat = ev._buf[B][0]
    lib = C.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "alphazero_amd/lib/dev/f32stamp.so"))
    lib.az_tower_forward_f32.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 10 + [C.c_int32, C.c_void_p]
    P = lambda t: C.c_void_p(t.data_ptr())
    if os.environ.get("TIME"):  # launch time only (any build of the library): TIME=1 python scripts/dev/f32_stamp.py path/to/lib.so
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get("REPS", "60"))
        for i in range(reps + 5):
            if i == 5:
                e0.record()
            lib.az_tower_forward_f32(N, 64, B, ev.n_layers, P(x), None, None, None, P(ev.w0), P(ev.wl), P(ev.bias), P(ev.headw), P(ev.headb),
                                     P(feat), ev.feat_stride, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record()
        torch.cuda.synchronize()
        print(f"{sys.argv[1] if len(sys.argv) > 1 else 'f32stamp'}: {e0.elapsed_time(e1) / reps:.3f} ms per launch of {B} boards")
        sys.exit(0)
    for _ in range(int(os.environ.get("REPS", "60"))):
        lib.az_tower_forward_f32(15, 64, B, ev.n_layers, P(x), None, None, None, P(ev.w0), P(ev.wl), P(ev.bias), P(ev.headw), P(ev.headb),
                                 P(feat), ev.feat_stride, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 128)()
    assert lib.az_debug_f32_stamps(out) == 0
    names = {1: "ticket+bar", 2: "ring+expand+bar", 3: "stem conv", 4: "stem bar+epi+bar"}
    for l in range(0, ev.n_layers, 2):
        names.update({5 + 4 * l: f"L{l} res+init", 6 + 4 * l: f"L{l} conv", 7 + 4 * l: f"L{l} bar", 8 + 4 * l: f"L{l} epi+bar",
                      9 + 4 * l: f"L{l+1} init", 10 + 4 * l: f"L{l+1} conv", 11 + 4 * l: f"L{l+1} bar", 12 + 4 * l: f"L{l+1} epi+bar"})
    names.update({26: "heads fma", 27: "heads shfl", 28: "heads c224+hp", 29: "heads bar", 30: "heads store"})
    for w in range(4):
        row = [int(out[w * 32 + k]) for k in range(32)]
        prev, parts = row[0], []
        for k in sorted(names):
            if row[k]:
                parts.append(f"{names[k]}={row[k] - prev}")
                prev = row[k]
        print(f"wave {w}: total={prev - row[0]}  " + "  ".join(parts))
    wg = (C.c_ulonglong * 4096)()
    assert lib.az_debug_f32_wg(wg) == 0
    import numpy as np
    a = np.array(list(wg), dtype=np.int64).reshape(1024, 4)[:512]
    t0 = a[:, 0].min()
    life = (a[:, 1] - a[:, 0]) / 100.0
    print(f"workgroups: start spread {(a[:, 0].max() - t0) / 100:.1f} us; end first {(a[:, 1].min() - t0) / 100:.1f} last {(a[:, 1].max() - t0) / 100:.1f} us; "
          f"lifetime min {life.min():.0f} median {np.median(life):.0f} max {life.max():.0f} us; boards {a[:, 2].min()}..{a[:, 2].max()}")
    hw = a[:, 3]
    waveid, simd, cu, sh, se, xcc = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7, 0
    key = (se * 2 + sh) * 16 + cu
    print("wave slots of wave 0:", np.bincount(waveid.astype(int)), " distinct (se,sh,cu):", len(set(key.tolist())))
    per = life / np.maximum(a[:, 2], 1)
    print(f"us per board: min {per.min():.1f} median {np.median(per):.1f} max {per.max():.1f}")
