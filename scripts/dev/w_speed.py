"""Launch time of the Winograd f32 tower (az_tower_w.hip) beside the direct f32 tower on dense batches:
   python scripts/dev/w_speed.py [boards] [blocks]"""
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
    torch.manual_seed(0)
    net = GomokuNet(15, 64, blocks).eval().cuda()
    x = torch.randint(-1, 2, (B, 9, 15, 15), device="cuda").float()
    flops = B * 225 * 2.0 * 9 * (9 * 64 + 2 * blocks * 64 * 64)
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
        print(f"{prec}: {ms:.3f} ms per {B} boards = {flops / ms / 1e9:.0f} TFLOP/s of direct-convolution flops", flush=True)
