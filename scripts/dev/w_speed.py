"""Launch time of the Winograd f32 tower (az_tower_w.hip) beside the direct f32 tower on dense batches:
   python scripts/dev/w_speed.py [boards] [blocks] [board size]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    torch.manual_seed(0)
    net = GomokuNet(n, 64, blocks).eval().cuda()
    x = torch.randint(-1, 2, (B, 9, n, n), device="cuda").float()
    flops = B * n * n * 2.0 * 9 * (9 * 64 + 2 * blocks * 64 * 64)
    for prec in ("f32w", "f32"):
        ev = FusedTowerEvaluator(net, prec)
        xx = x.contiguous()
        ev.kernel_events = []
        for _ in range(3):
            ev(xx)
        torch.cuda.synchronize()
        ev.kernel_events = []
        for _ in range(10):
            ev(xx)
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev.kernel_events)[len(ev.kernel_events) // 2]
        print(f"{prec}: {ms:.3f} ms per {B} boards of {n}x{n} = {flops / ms / 1e9:.0f} TFLOP/s of direct-convolution flops", flush=True)
    import ctypes as C

    from alphazero_amd.engine import load_library
    lib = load_library()
    if hasattr(lib, "az_debug_w_stamps"):  # a -DAZ_W_STAMP=1 build: phase boundaries of one board (workgroup 7, its sixth), s_memtime ticks (shader-clock cycles on this part: a board's total matches launch time / boards per workgroup at 2.39 GHz)
        ev = FusedTowerEvaluator(net, "f32w")
        ev(x.contiguous())
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 256)()
        lib.az_debug_w_stamps.argtypes = [C.c_void_p]
        assert lib.az_debug_w_stamps(buf) == 0
        names = {0: "top", 1: "slot known", 2: "input words in", 3: "expanded, preload", 4: "stem conv", 5: "barrier", 6: "stem epilogue"}
        for l in range(0, 2 * blocks, 2):
            names.update({7 + 3 * l: f"conv {l}", 8 + 3 * l: "barrier", 9 + 3 * l: "epilogue (res)", 10 + 3 * l: f"conv {l + 1}", 11 + 3 * l: "barrier", 12 + 3 * l: "epilogue"})
        names.update({28: "barrier", 29: "heads", 30: "barrier", 31: "feature rows"})
        for w in (0, 3, 7):
            t = [buf[w * 32 + k] for k in range(32)]
            print(f"wave {w}: board total {t[31] - t[0]} cycles")
            prev = t[0]
            for k in sorted(names):
                if t[k]:
                    print(f"   {names[k]:>18}: +{t[k] - prev:6d} cycles")
                    prev = t[k]
