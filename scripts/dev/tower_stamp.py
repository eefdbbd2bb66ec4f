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
vfeat = ev._buf[B][0]
lib = C.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "alphazero_amd/lib/dev/stamp.so"))
lib.az_tower_forward.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 12 + [C.c_int32, C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(int(os.environ.get("REPS", "200"))):  # long enough for the clock to settle under load
    lib.az_tower_forward(15, ev.ch, B, ev.n_layers, P(x), None, None, None, None, None, P(ev.w0), P(ev.wl), P(ev.bias), P(ev.headw), P(ev.headb), P(vfeat), ev.feat_stride, C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
t = vfeat.view(-1)[:512].view(torch.int64).view(4, 32).cpu()
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
    if int(row[24]):
        cyc, rt2 = int(row[22]) - int(row[20]), int(row[23]) - int(row[21])
        print(f"        whole workgroup: {int(row[24])} boards, {cyc} cycles = {cyc / int(row[24]):.0f} per board, {rt2 / 100:.1f} us, {cyc / rt2 * 0.1:.2f} GHz")
    rt = int(row[31]) - int(row[30])
    if rt > 0:
        print(f"        in-kernel clock over this board: {(int(row[29]) - int(row[0])) / rt * 0.1:.2f} GHz (s_memtime / s_memrealtime x 100 MHz)")
    print(f"        stage-end wait+barrier inside convs: block0={int(row[10])} block1(cum)={int(row[18])}")

nwg = 512
w = vfeat.view(-1)[512:512 + nwg * 16].view(torch.int64).view(nwg, 4).cpu()
st, en, nb = w[:, 0], w[:, 1], w[:, 2]
ok = nb > 0
t0 = int(st[ok].min())
dur = (en - st)[ok].float() / 100
print(f"workgroups reporting: {int(ok.sum())}; start spread {(int(st[ok].max()) - t0) / 100:.1f} us; end: first {(int(en[ok].min()) - t0) / 100:.1f} us, last {(int(en[ok].max()) - t0) / 100:.1f} us")
print(f"workgroup lifetime us: min {float(dur.min()):.1f} median {float(dur.median()):.1f} max {float(dur.max()):.1f}; boards per workgroup {int(nb[ok].min())}..{int(nb[ok].max())}")
xcc = (w[:, 3] & 0xF)
life = (en - st).float() / 100
for xid in sorted(set(xcc.tolist())):
    m = xcc == xid
    print(f"  XCC {xid}: {int(m.sum())} workgroups, lifetime min {float(life[m].min()):.0f} median {float(life[m].median()):.0f} max {float(life[m].max()):.0f} us")
order = torch.argsort(life)
print("  slowest blocks:", [(int(i), int(xcc[i]), round(float(life[i]))) for i in order[-12:]])
print("  fastest blocks:", [(int(i), int(xcc[i]), round(float(life[i]))) for i in order[:6]])

hw = (w[:, 3] >> 8)
cu = ((hw >> 8) & 0xF).tolist(); sh = ((hw >> 12) & 1).tolist(); se = ((hw >> 13) & 7).tolist(); simd = ((hw >> 4) & 3).tolist()
from collections import defaultdict
percu = defaultdict(list)
for i in range(nwg):
    percu[(int(xcc[i]), se[i], sh[i], cu[i])].append((round(float(life[i])), i, simd[i]))
print("distinct CUs:", len(percu), " workgroups per CU histogram:", sorted(__import__('collections').Counter(len(v) for v in percu.values()).items()))
slow = [k for k, v in percu.items() if max(x[0] for x in v) > 900]
print("CUs with a slow workgroup:", len(slow))
for k in sorted(percu)[:6] + slow[:8]:
    print("  ", k, percu[k])
