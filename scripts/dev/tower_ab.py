"""A/B timing of az_tower_forward across several builds of the library, in one process, interleaved.
  python scripts/dev/tower_ab.py lib1.so lib2.so ...     (paths relative to the repo root)
Times the tower kernel alone (HIP events on the launch stream), dense random +-1/0 input, 16,384 boards."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alphazero_amd.fused import FusedTowerEvaluator
from alphazero_amd.nets import GomokuNet

board = int(os.environ.get("BOARD", "15"))
B = int(os.environ.get("BATCH", "16384"))
blocks = int(os.environ.get("BLOCKS", "2"))
torch.manual_seed(0)
net = GomokuNet(board, int(os.environ.get("CHANNELS", "64")), blocks).eval().cuda()
ev = FusedTowerEvaluator(net)
x = torch.randint(-1, 2, (B, 9, board, board), device="cuda").to(torch.bfloat16)
if os.environ.get("ZERO"):  # all-zero weights and inputs: same instruction stream, minimal switching power
    ev.w0.zero_(); ev.wl.zero_(); x.zero_()
ref_p, ref_v = ev(x)
ref_p, ref_v = ref_p.clone(), ref_v.clone()
pfeat = ev._buf[B][0]
ref_pf = pfeat.float().clone()
P = lambda t: C.c_void_p(t.data_ptr())
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, path))
    lib.az_tower_forward.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 12 + [C.c_int32, C.c_void_p]
    libs.append((path, lib))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(lib):
    rc = lib.az_tower_forward(board, ev.ch, B, ev.n_layers, P(x), None, None, None, None, None, P(ev.w0), P(ev.wl), P(ev.bias),
                              P(ev.headw), P(ev.headb), P(pfeat), ev.feat_stride, st)
    assert rc == 0, rc


res = {p: [] for p, _ in libs}
for p, lib in libs:
    pfeat.zero_()
    run(lib)
    torch.cuda.synchronize()
    d = float((pfeat.float() - ref_pf).abs().max())
    print(f"{p}: max |pfeat - default lib| = {d:.3g}")
for rep in range(5):
    for p, lib in libs:
        for _ in range(3):
            run(lib)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 30
        for _ in range(n):
            run(lib)
        b.record()
        torch.cuda.synchronize()
        res[p].append(a.elapsed_time(b) / n)
fl = net.flops_per_eval() * B
for p, _ in libs:
    ms = sorted(res[p])[len(res[p]) // 2]
    print(f"{p}: {ms:.4f} ms  ({fl / ms / 1e9:.0f} TFLOP/s useful)  all={['%.4f' % v for v in res[p]]}")
