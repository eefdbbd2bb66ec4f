"""Print the per-phase cycle stamps of the -DAZ_TOWER_STAMP=1 build (alphazero_amd/lib/dev/stamp.so)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alphazero_amd.fused import FusedTowerEvaluator
from alphazero_amd.nets import GomokuNet
torch.manual_seed(0)
B = 16384
net = GomokuNet(15, 64, 2).eval().cuda()
ev = FusedTowerEvaluator(net)
x = torch.randint(-1, 2, (B, 9, 15, 15), device="cuda").to(torch.bfloat16)
ev(x)
pfeat, vfeat = ev._buf[B]
lib = C.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "alphazero_amd/lib/dev/stamp.so"))
lib.az_tower_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 13
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(5):
    lib.az_tower_forward(15, B, ev.n_layers, P(x), None, None, None, None, P(ev.w0), P(ev.wl), P(ev.bias), P(ev.headw), P(ev.headb), P(pfeat), P(vfeat), C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
t = vfeat.view(-1)[:256].view(torch.int64).view(4, 32).cpu()
names = {0: "start", 1: "input+bar", 2: "stem conv", 3: "stem epi+bar", 28: "heads", 29: "final bar"}
for b in range(2):
    k = 4 + 8 * b
    names.update({k: f"b{b} res+init", k + 1: f"b{b} conv1", k + 2: f"b{b} epi1+bar", k + 3: f"b{b} init2", k + 4: f"b{b} conv2", k + 5: f"b{b} epi2+bar"})
order = [0, 1, 2, 3] + [4 + 8 * b + i for b in range(2) for i in range(6)] + [28, 29]
for w in range(4):
    row = t[w]
    out = []
    prev = int(row[0])
    for k in order[1:]:
        out.append(f"{names[k]}={int(row[k]) - prev}")
        prev = int(row[k])
    print(f"wave {w}: total={int(row[29]) - int(row[0])}  " + "  ".join(out))
    print(f"        stage-end wait+barrier inside convs: block0={int(row[10])} block1(cum)={int(row[18])}")
