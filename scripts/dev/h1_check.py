"""Accuracy (vs an f64 evaluation) and launch time of the row-tile tower (az_tower_h1.hip: "f16", "bf16r") beside the other
towers on dense 15x15 batches:   python scripts/dev/h1_check.py [boards] [blocks]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet
    from tests.test_gpu_tower import _randomised

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f16", "bf16r", "bf16", "f16x2"]
    torch.manual_seed(0)
    net = _randomised(GomokuNet(15, 64, blocks)).eval().cuda()
    x = torch.randint(-1, 2, (B, 9, 15, 15), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(B, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    nref = min(B, 512)
    with torch.inference_mode():
        rp, rv = net.double()(x[:nref].double())
    net.float()
    scale = float(rp.abs().max())
    flops = B * 225 * 2.0 * 9 * (9 * 64 + 2 * blocks * 64 * 64)
    for prec in precs:
        ev = FusedTowerEvaluator(net, prec)
        xx = x.bfloat16().contiguous() if prec == "bf16" else x.contiguous()
        p, v = ev(xx)
        torch.cuda.synchronize()
        err = float((p[:nref].double() - rp).abs().max())
        errm = float((p[:nref].double() - rp).abs().mean())
        verr = float((v[:nref].double() - rv).abs().max())
        p2, _ = ev(xx)
        same = bool(torch.equal(p, p2))
        ev.kernel_events = []
        for _ in range(3):
            ev(xx)
        torch.cuda.synchronize()
        ev.kernel_events = []
        for _ in range(10):
            ev(xx)
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev.kernel_events)[len(ev.kernel_events) // 2]
        print(f"{prec}: max |dlogit| {err:.3e} mean {errm:.3e} (scale {scale:.2f}) max |dvalue| {verr:.3e} repeatable {same} overflow {ev.overflowed()} | "
              f"{ms:.3f} ms per {B} boards = {flops / ms / 1e9:.0f} TFLOP/s", flush=True)
